"""Per-way contrastive loss (reference models/mpti.py:226-313) and the training-only debug metrics
(mpti.py:515-568) as autograd edges around r3d_contrast_fwd / r3d_contrast_bwd; batches of E episodes as in
train_ops.py (``ctx.E``, ``ctx.ep_rows`` set by the caller; E = 1 by default)."""
import torch

from . import _lib, ops
from .ops import _p, _st


class ContrastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sfeat, W, b, model, support_y, support_flag):
        """sfeat: support rows of episode 0 (episode e's start ctx.ep_rows rows further on) -> loss (E,) (E = 1: 0-d)."""
        lib = _lib.load()
        E, ep_rows = getattr(ctx, "E", 1), getattr(ctx, "ep_rows", 0)
        S, N, D = model.n_way * model.k_shot, model.n_points, model.feat_dim
        dev = sfeat.device
        words = lib.r3d_contrast_ws_words(model.n_way, model.k_shot, N)
        ws = torch.empty(E, words, device=dev, dtype=torch.float32)
        loss = torch.empty(E, device=dev, dtype=torch.float32)
        sy = support_y.reshape(E, S, N).to(torch.int32).contiguous()
        sf = support_flag.reshape(E, S).to(torch.int32).contiguous()
        Wc, bc = W.detach().contiguous(), b.detach().contiguous()
        _lib.check(lib.r3d_contrast_fwd_batched(E, _p(sfeat), sfeat.stride(0), ep_rows, D, _p(sy), _p(sf), model.n_way,
                                                model.k_shot, N, _p(Wc), _p(bc), 0.1, _p(loss), _p(ws), words, words, _st()))
        ctx.ws, ctx.model, ctx.shape, ctx.E, ctx.ep_rows, ctx.words = ws, model, sfeat.shape, E, ep_rows, words
        return loss[0] if E == 1 else loss

    @staticmethod
    def backward(ctx, gloss):
        """-> (dfeat, dW, db): dfeat has the shape of sfeat (E = 1) or of the batch's whole feature matrix (E > 1; query
        rows zero); dW, db are summed over the batch."""
        model = ctx.model
        lib = _lib.load()
        dev = ctx.ws.device
        D, N = model.feat_dim, model.n_points
        E, ep_rows = ctx.E, ctx.ep_rows
        gs = gloss.reshape(-1)[:1].to(torch.float32).contiguous()
        shape = ctx.shape if E == 1 else (E * ep_rows, D)
        dfeat = torch.zeros(shape, device=dev, dtype=torch.float32)
        dW = torch.empty(128, D, device=dev, dtype=torch.float32)
        db = torch.empty(128, device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_contrast_bwd_batched(E, D, model.n_way, model.k_shot, N, _p(gs), _p(dfeat), D, ep_rows, _p(dW), _p(db),
                                                _p(ctx.ws), ctx.words, _st()))
        return dfeat, dW, db, None, None, None


def per_way_contrast_loss(model, sfeat, support_y, support_flag):
    return ContrastFn.apply(sfeat, model.proj.weight, model.proj.bias, model, support_y, support_flag)


def train_debug_metrics(model, support_y, gt_support_y, query_y, gt_query_y, logger, E=None):
    """(query_acc_LP, query_acc_original, clean_ratio_LP_avg, clean_ratio_original_avg) as 0-d device tensors; with E
    given: one row per episode of the batch the head buffers hold, as an (E, 4) tensor."""
    hb = model._head[1]
    lib = _lib.load()
    N = model.n_points
    S = model.n_way * model.k_shot
    dev = hb.Z.device
    n_ep = hb.E
    assert E is None or E == n_ep
    out = torch.empty(n_ep, 4, device=dev, dtype=torch.float32)
    pred = model._train_pred
    n_qpts = pred.numel() // n_ep
    qy = query_y.to(torch.int64).contiguous()
    gq = (gt_query_y if gt_query_y is not None else query_y).to(torch.int64).contiguous()
    gs = (gt_support_y if gt_support_y is not None else support_y).reshape(-1).to(torch.int32).contiguous()
    _lib.check(lib.r3d_train_metrics_batched(n_ep, _p(pred), _p(qy), _p(gq), n_qpts, _p(hb.Z), hb.n_cap, _p(hb.desc), 32,
                                             _p(hb.proto_ws), hb.proto_stride, _p(hb.assign), 2 * S * N, _p(gs), model.n_way,
                                             model.k_shot, N, _p(out), _st()))
    if E is not None:
        return out
    if logger is not None:  # the reference prints these every step (mpti.py:546,568): a host sync, as there
        v = out[0].tolist()
        logger.cprint('after label propagation: QUERY prediction acc: {:.3f}, original_acc: {:.3f}'.format(v[0], v[1]))
        logger.cprint('after label propagation: clean_ratio_LP: {:.3f}, clean_ratio_original: {:.3f}'.format(v[2], v[3]))
    return out[0, 0], out[0, 1], out[0, 2], out[0, 3]
