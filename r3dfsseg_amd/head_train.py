"""Training-mode forward of MPTI_SelfAtten (reference models/mpti.py:414-577 with train=True) and the
autograd edge of the transductive head.  Compute lives in libr3d_hip.so; this file orders launches."""
import torch

from . import _lib, ops, train_ops as T
from .ops import _p, _st, _timed


class HeadLPFn(torch.autograd.Function):
    """(sfeat (S*N,192), qfeat (n_q*N,192)) -> lp_loss; also leaves logits / Z in the head buffers."""

    @staticmethod
    def forward(ctx, sfeat, qfeat, model, support_y, query_y):
        S, N = model.n_way * model.k_shot, model.n_points
        n_q = qfeat.shape[0] // N
        hb = model._head_buffers(n_q, sfeat.device)
        sfeatT = ops.pm_to_cm(sfeat, S, N)
        sy = support_y.reshape(S, N).to(torch.int32).contiguous()
        ops.head_prototypes(hb, sy, None, sfeat, sfeatT, qfeat)
        nbr = ops.knn(hb.nodes, 1, hb.n_cap, hb.kp1, mode=ops.SCORE_L2, n_valid=hb.desc[ops.HD_N_NODES:],
                      status=hb.knn_status)
        # same launch-budget policy as eval (mpti.py: _lp_next_budget); the training loop's own host sync
        # (loss.item(), mpti_train_noise.py:107) is where lp_converged() can be checked
        ctx.budget = model._lp_next_budget()
        ops.label_propagate(hb, nbr, model.sigma, 0.99, ctx.budget, model.lp_tol)
        model._lp_post(hb)
        labels = query_y.to(torch.int64).contiguous()
        logits, loss, pred = ops.query_logits_ce(hb, n_q, model.n_classes, labels)
        ctx.model, ctx.hb, ctx.labels, ctx.n_q = model, hb, labels, n_q
        ctx.shapes = (sfeat.shape, qfeat.shape)
        model._train_logits, model._train_pred = logits, pred
        return loss

    @staticmethod
    def backward(ctx, gloss):
        model, hb, labels, n_q = ctx.model, ctx.hb, ctx.labels, ctx.n_q
        lib = _lib.load()
        dev = hb.Z.device
        N, D = model.n_points, model.feat_dim
        gs = gloss.reshape(1).to(torch.float32).contiguous()
        G = torch.empty(hb.n_cap, 4, device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_ce_grad(_p(hb.Z), _p(hb.desc[ops.HD_N_PROTO:]), hb.n_cap, n_q * N, model.n_classes, _p(labels),
                                   _p(gs), _p(G), _st()))
        lam = torch.empty(hb.n_cap, 4, device=dev, dtype=torch.float32)
        dnodes = torch.empty(hb.n_cap, D, device=dev, dtype=torch.float32)
        with _timed("label_propagate_bwd"):
          _lib.check(lib.r3d_label_propagate_bwd(_p(hb.nodes), hb.nodes.stride(0), D, hb.kp1, _p(hb.Z), _p(G),
                                               _p(hb.desc[ops.HD_N_NODES:]), hb.n_cap, float(model.sigma), 0.99,
                                               int(min(model.lp_max_iter, 2 * ctx.budget)), float(model.lp_tol), _p(lam),
                                               _p(dnodes), D,
                                               _p(hb.lp_ws), _p(hb.stats_bwd), _st()))
        dsfeat = torch.zeros(ctx.shapes[0], device=dev, dtype=torch.float32)
        dqfeat = torch.empty(ctx.shapes[1], device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_head_prototypes_bwd(_p(dnodes), D, model.n_way, model.k_shot, N, D, n_q * N, _p(hb.desc),
                                               _p(hb.assign), _p(hb.cluster_count), _p(hb.proto_ws), _p(dsfeat), D,
                                               _p(dqfeat), D, _st()))
        return dsfeat, dqfeat, None, None, None


def mpti_train_forward(model, support_x, support_y, query_x, query_y, gt_support_y, gt_query_y, logger, support_flag):
    """Returns the reference's 7-tuple (mpti.py:573-575): query_pred, lp_loss, contrast_loss, query_acc_LP,
    query_acc_original, clean_ratio_LP_avg, clean_ratio_original_avg."""
    from . import contrast
    S, N = model.n_way * model.k_shot, model.n_points
    model._drop_seed = getattr(model, "_drop_seed", 0) + 2
    sx = support_x.reshape(S, model.in_channels, N)
    # two getFeatures calls, each with its own BatchNorm batch statistics (mpti.py:434,436)
    sfeat = T.get_features_train(model, sx, model._drop_seed)
    qfeat = T.get_features_train(model, query_x, model._drop_seed + 1)
    contrast_loss = contrast.per_way_contrast_loss(model, sfeat, support_y, support_flag)
    lp_loss = HeadLPFn.apply(sfeat, qfeat, model, support_y, query_y)
    logits = model._train_logits
    metrics = contrast.train_debug_metrics(model, support_y, gt_support_y, query_y, gt_query_y, logger)
    return (logits, lp_loss, contrast_loss) + metrics
