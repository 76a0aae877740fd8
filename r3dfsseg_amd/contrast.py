"""Per-way contrastive loss (reference models/mpti.py:226-313) and the training-only debug metrics
(mpti.py:515-568) as autograd edges around r3d_contrast_fwd / r3d_contrast_bwd."""
import torch

from . import _lib, ops
from .ops import _p, _st


class ContrastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sfeat, W, b, model, support_y, support_flag):
        lib = _lib.load()
        S, N, D = model.n_way * model.k_shot, model.n_points, model.feat_dim
        dev = sfeat.device
        ws = torch.empty(lib.r3d_contrast_ws_words(model.n_way, model.k_shot, N), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        sy = support_y.reshape(S, N).to(torch.int32).contiguous()
        sf = support_flag.reshape(S).to(torch.int32).contiguous()
        Wc, bc = W.detach().contiguous(), b.detach().contiguous()
        _lib.check(lib.r3d_contrast_fwd(_p(sfeat), sfeat.stride(0), D, _p(sy), _p(sf), model.n_way, model.k_shot, N, _p(Wc),
                                        _p(bc), 0.1, _p(loss), _p(ws), ws.numel(), _st()))
        ctx.ws, ctx.model, ctx.shape = ws, model, sfeat.shape
        return loss

    @staticmethod
    def backward(ctx, gloss):
        model = ctx.model
        lib = _lib.load()
        dev = ctx.ws.device
        D, N = model.feat_dim, model.n_points
        gs = gloss.reshape(1).to(torch.float32).contiguous()
        dfeat = torch.zeros(ctx.shape, device=dev, dtype=torch.float32)
        dW = torch.empty(128, D, device=dev, dtype=torch.float32)
        db = torch.empty(128, device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_contrast_bwd(D, model.n_way, model.k_shot, N, _p(gs), _p(dfeat), D, _p(dW), _p(db), _p(ctx.ws),
                                        _st()))
        return dfeat, dW, db, None, None, None


def per_way_contrast_loss(model, sfeat, support_y, support_flag):
    return ContrastFn.apply(sfeat, model.proj.weight, model.proj.bias, model, support_y, support_flag)


def train_debug_metrics(model, support_y, gt_support_y, query_y, gt_query_y, logger):
    """(query_acc_LP, query_acc_original, clean_ratio_LP_avg, clean_ratio_original_avg) as 0-d device tensors."""
    hb = model._head[1]
    lib = _lib.load()
    N = model.n_points
    dev = hb.Z.device
    out = torch.empty(4, device=dev, dtype=torch.float32)
    pred = model._train_pred
    qy = query_y.to(torch.int64).contiguous()
    gq = (gt_query_y if gt_query_y is not None else query_y).to(torch.int64).contiguous()
    gs = (gt_support_y if gt_support_y is not None else support_y).reshape(-1).to(torch.int32).contiguous()
    _lib.check(lib.r3d_train_metrics(_p(pred), _p(qy), _p(gq), pred.numel(), _p(hb.Z), _p(hb.desc), _p(hb.proto_ws),
                                     _p(hb.assign), _p(gs), model.n_way, model.k_shot, N, _p(out), _st()))
    if logger is not None:  # the reference prints these every step (mpti.py:546,568): a host sync, as there
        v = out.tolist()
        logger.cprint('after label propagation: QUERY prediction acc: {:.3f}, original_acc: {:.3f}'.format(v[0], v[1]))
        logger.cprint('after label propagation: clean_ratio_LP: {:.3f}, clean_ratio_original: {:.3f}'.format(v[2], v[3]))
    return out[0], out[1], out[2], out[3]
