"""Host-side mirror of the reference's models/proto_learner.py::ProtoLearner (test path; the
reference's own train() cannot run, see SURVEY.md "facts that contradict")."""
import torch
from torch import optim

from .checkpoint_util import load_model_checkpoint, load_pretrain_checkpoint
from .protonet import ProtoNet


class ProtoLearner(object):
    def __init__(self, args, mode='train'):
        self.model = ProtoNet(args)
        if not torch.cuda.is_available():
            raise RuntimeError("ProtoLearner needs an MI355X: the forward pass has no CPU path")
        self.model.cuda()
        synthetic = 'synthetic' in (getattr(args, 'pretrain_checkpoint_path', None), getattr(args, 'model_checkpoint_path', None))
        if synthetic:
            from . import synthetic as S
            sd = S.make_state_dict(vars(args))
            self.model.load_state_dict({k: v for k, v in sd.items() if not k.startswith('proj.')})
        if mode == 'train':
            head = self.model.att_learner if args.use_attention else self.model.linear_mapper
            self.optimizer = torch.optim.Adam(
                [{'params': self.model.encoder.parameters(), 'lr': 0.0001},
                 {'params': self.model.base_learner.parameters()},
                 {'params': head.parameters()}], lr=args.lr)
            self.lr_scheduler = optim.lr_scheduler.StepLR(self.optimizer, step_size=args.step_size, gamma=args.gamma)
            if not synthetic:
                self.model = load_pretrain_checkpoint(self.model, args.pretrain_checkpoint_path)
        elif mode == 'test':
            if not synthetic:
                self.model = load_model_checkpoint(self.model, args.model_checkpoint_path, mode='test')
        else:
            raise ValueError('Wrong GMMLearner mode (%s)! Option:train/test' % mode)

    def train(self, data, logger):
        raise NotImplementedError("the reference's ProtoLearner.train is broken (proto_learner.py:57)")

    def test(self, data, sampled_classes, step=None, path=None):
        [support_x, support_y, query_x, query_y, _, _, gt_support_y] = data
        self.model.eval()
        with torch.no_grad():
            logits, loss = self.model(support_x, support_y, query_x, query_y)
            pred = logits.argmax(dim=1)
            correct = torch.eq(pred, query_y).sum().item()
            accuracy = correct / (query_y.shape[0] * query_y.shape[1])
        return pred, loss, accuracy
