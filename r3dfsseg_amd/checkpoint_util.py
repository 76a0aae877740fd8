"""Checkpoint interop with the reference (utils/checkpoint_util.py:9-50): same file layout
({'params': encoder.state_dict()} for pre-training, {'iteration','model_state_dict',
'optimizer_state_dict','loss','IoU'} for training) and the same key names."""
import os

import torch


def load_pretrain_checkpoint(model, pretrain_checkpoint_path):
    model_dict = model.state_dict()
    if pretrain_checkpoint_path is not None:
        print('Load encoder module from pretrained checkpoint...')
        pretrained_dict = torch.load(pretrain_checkpoint_path, map_location='cpu')['params']
        pretrained_dict = {'encoder.' + k: v for k, v in pretrained_dict.items()}
        pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in model_dict}
        model_dict.update(pretrained_dict)
        model.load_state_dict(model_dict)
    else:
        raise ValueError('Pretrained checkpoint must be given.')
    return model


def load_model_checkpoint(model, model_checkpoint_path, optimizer=None, mode='test'):
    try:
        checkpoint = torch.load(os.path.join(model_checkpoint_path, 'checkpoint.tar'), map_location='cpu')
        start_iter = checkpoint['iteration']
        start_iou = checkpoint['IoU']
    except Exception:
        raise ValueError('Model checkpoint file must be correctly given (%s).' % model_checkpoint_path)
    model.load_state_dict(checkpoint['model_state_dict'], strict=False)
    if mode == 'test':
        print('Load model checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
        return model
    try:
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    except Exception:
        print('Checkpoint does not include optimizer state dict...')
    print('Resume from checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
    return model, optimizer


def save_pretrain_checkpoint(model, output_path, epoch=None):
    name = 'checkpoint_{}.tar'.format(epoch) if epoch is not None else 'checkpoint.tar'
    torch.save(dict(params=model.encoder.state_dict()), os.path.join(output_path, name))
