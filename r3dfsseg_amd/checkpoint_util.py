"""Checkpoint files of the reference, read and written for the MI355X modules.

File layouts (reference: utils/checkpoint_util.py:9-50 reads them, mpti_train_noise.py:135-152 writes them):
  * training checkpoint  <dir>/checkpoint.tar (best) or <dir>/checkpoint_<iteration>.tar:
        {'iteration', 'model_state_dict', 'optimizer_state_dict', 'loss', 'IoU'}
  * pre-training checkpoint (attMPTI's supervised stage):  {'params': encoder.state_dict()} with keys relative to
    the DGCNN encoder; only the keys the few-shot model also has are taken (a pre-training segmentor carries more).
Entry points keep the reference's names, argument order and return values so that MPTILearner_V3 / ProtoLearner
construct exactly as there.  What is specific to this build: tensors are shape-checked against the module before
anything is copied (a HIP kernel reading a wrongly shaped weight is a memory fault, not a Python error), copies go
INTO the existing parameter storage (captured episode hipGraphs hold raw pointers), and the BatchNorm-folded
weight caches the kernels read are invalidated so that the next forward re-derives them.
"""
import os
import pickle

import torch

TRAIN_KEYS = ('iteration', 'model_state_dict', 'optimizer_state_dict', 'loss', 'IoU')


def _numpy_scalar_globals():
    """Allow-list for the restricted unpickler: the reference writes 'IoU' / 'loss' as numpy scalars
    (mpti_train_noise.py:135-152), which pickle as numpy's `scalar` reconstructor plus a dtype -- data, no code.  Files
    written under numpy < 2 name the reconstructor numpy.core.multiarray.scalar, newer ones numpy._core.multiarray.scalar."""
    import numpy as np
    try:
        import numpy._core.multiarray as ncm   # numpy >= 1.26 / 2.x
    except ImportError:
        import numpy.core.multiarray as ncm    # older numpy: the only name it pickles under
    allowed = [ncm.scalar, (ncm.scalar, 'numpy.core.multiarray.scalar'), (ncm.scalar, 'numpy._core.multiarray.scalar'), np.dtype]
    for t in (np.float64, np.float32, np.float16, np.int64, np.int32, np.bool_):
        allowed.append(type(np.dtype(t)))
    return allowed


def _read(path):
    """torch.load with the restricted unpickler (tensors, containers and numpy scalars only).  A file that needs
    anything else is refused -- pre-trained checkpoints of this model are passed around between groups, and a pickle
    can run code -- unless R3D_TRUST_CHECKPOINTS=1 explicitly opts into the full unpickler (logged)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    try:
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as err:
        if os.environ.get('R3D_TRUST_CHECKPOINTS') != '1':
            raise RuntimeError('%s needs more than tensors and numpy scalars to unpickle (%s); set '
                               'R3D_TRUST_CHECKPOINTS=1 only if you trust where the file came from' % (
                                   path, str(err).splitlines()[0]))
        print('WARNING: %s loaded with the unrestricted unpickler (R3D_TRUST_CHECKPOINTS=1)' % path)
        return torch.load(path, map_location='cpu', weights_only=False)


def _assign(model, tensors, what):
    """Copy `tensors` (name -> tensor) into the module's parameters / buffers in place.  Returns (loaded, skipped)
    name lists; a name the module has with another shape is an error."""
    own = model.state_dict(keep_vars=True)
    loaded, skipped = [], []
    with torch.no_grad():
        for name, value in tensors.items():
            dst = own.get(name)
            if dst is None:
                skipped.append(name)
                continue
            if tuple(dst.shape) != tuple(value.shape):
                raise ValueError('%s: %s has shape %s in the file, %s in the model' % (
                    what, name, tuple(value.shape), tuple(dst.shape)))
            dst.copy_(value.to(dst.dtype))
            loaded.append(name)
    for mod in model.modules():  # BatchNorm-folded / fused copies are derived data
        if hasattr(mod, '_folded') and mod._folded is not None:
            mod._folded = (None, mod._folded[1])  # keep the storage (graphs point at it), drop the validity key
    return loaded, skipped


def load_pretrain_checkpoint(model, pretrain_checkpoint_path):
    """utils/checkpoint_util.py:9-23: encoder weights of the supervised pre-training stage."""
    if pretrain_checkpoint_path is None:
        raise ValueError('Pretrained checkpoint must be given.')
    blob = _read(pretrain_checkpoint_path)
    if not isinstance(blob, dict) or 'params' not in blob:
        raise ValueError('%s is not a pre-training checkpoint (no "params" entry)' % pretrain_checkpoint_path)
    loaded, skipped = _assign(model, {'encoder.' + k: v for k, v in blob['params'].items()}, pretrain_checkpoint_path)
    if not loaded:
        raise ValueError('%s holds no tensor of this encoder' % pretrain_checkpoint_path)
    print('Load encoder module from pretrained checkpoint... (%d tensors, %d not in this model)' % (
        len(loaded), len(skipped)))
    return model


def load_model_checkpoint(model, model_checkpoint_path, optimizer=None, mode='test'):
    """utils/checkpoint_util.py:26-44: <dir>/checkpoint.tar; non-strict on the model, optimizer state best effort.
    Returns model (mode 'test') or (model, optimizer)."""
    path = os.path.join(str(model_checkpoint_path), 'checkpoint.tar')
    try:
        ckpt = _read(path)
        start_iter, start_iou = ckpt['iteration'], ckpt['IoU']
        state = ckpt['model_state_dict']
    except (OSError, KeyError, TypeError, RuntimeError, EOFError):
        raise ValueError('Model checkpoint file must be correctly given (%s).' % model_checkpoint_path)
    loaded, skipped = _assign(model, state, path)
    missing = [k for k in model.state_dict() if k not in state]
    model.checkpoint_meta = dict(iteration=int(start_iter), IoU=float(start_iou), loss=ckpt.get('loss'),
                                 missing=missing, unexpected=skipped)
    if mode == 'test':
        print('Load model checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
        return model
    restored = False
    if optimizer is not None and ckpt.get('optimizer_state_dict') is not None:
        try:
            optimizer.load_state_dict(ckpt['optimizer_state_dict'])
            restored = True
        except (ValueError, KeyError, RuntimeError):
            pass
    if not restored:
        print('Checkpoint does not include optimizer state dict...')
    print('Resume from checkpoint at Iteration %d (IoU %f)...' % (start_iter, start_iou))
    return model, optimizer


def save_model_checkpoint(learner, output_dir, iteration, loss, iou, best=True):
    """What mpti_train_noise.py:135-152 writes: checkpoint.tar (best so far) or checkpoint_<iteration>.tar."""
    from . import dist as D
    D.warn_rank_local_stats(learner.model, 'checkpoint written')
    name = 'checkpoint.tar' if best else 'checkpoint_%d.tar' % iteration
    path = os.path.join(output_dir, name)
    torch.save(dict(zip(TRAIN_KEYS, (iteration, learner.model.state_dict(), learner.optimizer.state_dict(), loss, iou))),
               path)
    return path


def save_pretrain_checkpoint(model, output_path, epoch=None):
    """utils/checkpoint_util.py:47-50."""
    name = 'checkpoint.tar' if epoch is None else 'checkpoint_%s.tar' % epoch
    path = os.path.join(output_path, name)
    torch.save({'params': model.encoder.state_dict()}, path)
    return path
