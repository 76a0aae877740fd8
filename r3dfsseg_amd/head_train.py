"""Training-mode forward of MPTI_SelfAtten (reference models/mpti.py:414-577 with train=True) and the
autograd edge of the transductive head.  Compute lives in libr3d_hip.so; this file orders launches.

Like train_ops.py, everything here runs on a batch of E episodes (ops.SegLayout); E = 1 is the reference's schedule."""
from types import SimpleNamespace

import torch

from . import _lib, ops, train_ops as T
from .ops import SegLayout, _p, _st, _timed


class HeadLPFn(torch.autograd.Function):
    """(sfeat, qfeat) -> lp_loss (E,); also leaves logits / Z in the head buffers.

    sfeat / qfeat: the support / query rows of episode 0; with ``ctx.E > 1`` (set by the caller) they are views into ONE
    feature matrix in which episode e's rows start ``ctx.ep_rows`` rows further on, and backward returns the gradient
    of that whole matrix as its first result."""

    @staticmethod
    def forward(ctx, sfeat, qfeat, model, support_y, query_y):
        E = getattr(ctx, "E", 1)
        ep_rows = getattr(ctx, "ep_rows", 0)
        S, N = model.n_way * model.k_shot, model.n_points
        n_q = query_y.shape[-2]
        hb = model._head_buffers(n_q, sfeat.device, E)
        sy = support_y.reshape(E, S, N).to(torch.int32).contiguous()
        if model._lp_force:  # the conservative re-run (see MPTILearner_V3.train): one FPS launch per round as well
            hb.fps_one_launch = False
        ops.head_prototypes(hb, sy, None, sfeat, qfeat, ep_rows)
        nbr = ops.knn_nodes(hb, exact=model._lp_force)
        if model.nbr_patch is not None:  # parity tests only (mpti.MPTI_SelfAtten.nbr_patch)
            nbr = model.nbr_patch(nbr)
        if model._trace is not None:
            model._trace["nbr"] = nbr
        # same launch-budget policy as eval (mpti.py: _lp_next_budget); MPTILearner_V3.train / DPTrainer.step check
        # lp_converged(backward=True) before the optimiser step and redo the episode on this conservative schedule
        ctx.budget = model._lp_next_budget()
        ops.label_propagate(hb, nbr, model.sigma, 0.99, ctx.budget, model.lp_tol)
        model._lp_post(hb)
        labels = query_y.reshape(E, n_q, N).to(torch.int64).contiguous()
        logits, loss, pred = ops.query_logits_ce(hb, n_q, model.n_classes, labels)
        ctx.model, ctx.hb, ctx.labels, ctx.n_q, ctx.E, ctx.ep_rows = model, hb, labels, n_q, E, ep_rows
        ctx.shapes = (sfeat.shape, qfeat.shape)
        model._train_logits, model._train_pred = logits, pred
        return loss

    @staticmethod
    def backward(ctx, gloss):
        model, hb, labels, n_q, E, ep_rows = ctx.model, ctx.hb, ctx.labels, ctx.n_q, ctx.E, ctx.ep_rows
        lib = _lib.load()
        dev = hb.Z.device
        N, D = model.n_points, model.feat_dim
        S = model.n_way * model.k_shot
        gs = gloss.reshape(-1)[:1].to(torch.float32).contiguous()  # (the step's loss is the SUM over episodes: one scale)
        pl = E * hb.n_cap  # rows of one plane of label columns (ops.HeadBuffers: two planes for more than 3 ways)
        G = torch.empty(hb.planes * pl, 4, device=dev, dtype=torch.float32)
        _lib.check(lib.r3d_ce_grad_batched(E, _p(hb.Z), _p(hb.n_proto_ptr()), 32, hb.n_cap, n_q * N, model.n_classes, _p(labels),
                                           _p(gs), _p(G), _st()))
        lam = torch.empty(pl, 4, device=dev, dtype=torch.float32)
        dnodes = torch.empty(pl, D, device=dev, dtype=torch.float32)
        budget = int(min(model.lp_max_iter, ctx.budget + max(4, ctx.budget // 4)))
        with _timed("label_propagate_bwd"):
            for plane in range(hb.planes):  # the adjoint is column-wise independent as well: the planes' dnodes add
                dn = dnodes if plane == 0 else torch.empty_like(dnodes)
                stats = hb.stats_bwd if plane == 0 else torch.zeros(E, 2, device=dev, dtype=torch.int32)
                _lib.check(lib.r3d_label_propagate_bwd_batched(
                    E, _p(hb.nodes), hb.nodes.stride(0), D, hb.kp1, _p(hb.Z[plane * pl:]), _p(G[plane * pl:]),
                    _p(hb.n_nodes_ptr()), 32, hb.n_cap, float(model.sigma), 0.99, budget, float(model.lp_tol), _p(lam), _p(dn),
                    D, _p(hb.lp_ws), hb.lp_words, hb.lp_stride, _p(stats), 2, _st()))
                if plane:
                    dnodes.add_(dn)
                    sb = hb.stats_bwd.view(E, 2)
                    sb[:, 0] = torch.minimum(sb[:, 0], stats[:, 0])
                    sb[:, 1] = torch.maximum(sb[:, 1], stats[:, 1])
        # one buffer over the batch's rows, per episode support rows then query rows: the encoder backward takes it whole
        assert E == 1 or ep_rows == (S + n_q) * N
        rows = E * (S + n_q) * N
        dfeat = torch.empty(rows, D, device=dev, dtype=torch.float32)
        dfeat.view(E, (S + n_q) * N, D)[:, :S * N].zero_()
        dsfeat, dqfeat = dfeat, dfeat[S * N:]
        _lib.check(lib.r3d_head_prototypes_bwd_batched(E, _p(dnodes), D, hb.n_cap, model.n_way, model.k_shot, N, D, n_q * N,
                                                       _p(hb.desc), 32, _p(hb.assign), 2 * S * N, _p(hb.cluster_count), hb.n_cap,
                                                       _p(hb.proto_ws), hb.proto_stride, _p(dsfeat), D, ep_rows, _p(dqfeat), D,
                                                       ep_rows, _st()))
        ctx.dfeat_full = dfeat  # (explicit_train_batch takes the whole buffer)
        if E > 1:
            return dfeat, None, None, None, None
        return dfeat[:S * N], dqfeat, None, None, None


def mpti_train_forward(model, support_x, support_y, query_x, query_y, gt_support_y, gt_query_y, logger, support_flag):
    """Returns the reference's 7-tuple (mpti.py:573-575): query_pred, lp_loss, contrast_loss, query_acc_LP,
    query_acc_original, clean_ratio_LP_avg, clean_ratio_original_avg."""
    from . import contrast
    S, N = model.n_way * model.k_shot, model.n_points
    slot = model._slot
    if slot.seed_dev is not None:  # captured launch sequence: the seed advances in device memory
        slot.seed_dev.add_(2)
        seed = 0
    else:
        model._drop_seed = getattr(model, "_drop_seed", 0) + 2
        seed = model._drop_seed
    sx = support_x.reshape(S, model.in_channels, N)
    # two getFeatures calls, each with its own BatchNorm batch statistics (mpti.py:434,436), through one launch sequence
    # over the S + Q clouds
    seg = SegLayout(1, S, query_x.shape[0], N)
    sfeat, qfeat = T.get_features_train(model, ops.cat_clouds(sx, query_x, 0), seed, seg=seg)
    if model._trace is not None:  # parity tests read the features and, after backward(), their gradients
        sfeat.retain_grad()
        qfeat.retain_grad()
        model._trace.update(sfeat=sfeat, qfeat=qfeat)
    contrast_loss = contrast.per_way_contrast_loss(model, sfeat, support_y, support_flag)
    lp_loss = HeadLPFn.apply(sfeat, qfeat, model, support_y, query_y)
    logits = model._train_logits
    metrics = contrast.train_debug_metrics(model, support_y, gt_support_y, query_y, gt_query_y, logger)
    model._last_train_parts = (lp_loss.detach(), contrast_loss.detach(), metrics)
    return (logits, lp_loss, contrast_loss) + metrics


def explicit_train_batch(model, batch, grad_sink, loss_weight=0.1):
    """Forward + backward of the E episodes of `batch` (batch.EpisodeBatch) as ONE fixed launch sequence without the
    autograd engine: the forward halves of the three autograd Functions run with plain namespaces as their ctx, then
    their backward halves in dependency order, and every parameter gradient -- summed over the E episodes where it is
    produced -- is ADDED into grad_sink[i] (views in the order of model.parameters(), requires_grad only).  Same
    kernels and, per episode, the same results as ``loss = lp + loss_weight * contrast; loss.backward()``
    (models/mpti_learner.py:66-68) episode after episode.  Returns (loss (E,), logits (E, n_q, n_classes, N),
    metrics (E, 4), lp_loss (E,), contrast_loss (E,))."""
    from . import contrast
    E = batch.E
    S, N = model.n_way * model.k_shot, model.n_points
    Q = batch.query_x.shape[1]
    with torch.no_grad():
        slot = model._slot
        if slot.seed_dev is not None:
            slot.seed_dev.add_(2 * E)
            seed = 2 - 2 * E  # episode e draws seed_dev + 2 - 2 E + 2 e: the values E single-episode sequences would
        else:
            seed = getattr(model, "_drop_seed", 0) + 2
            model._drop_seed = seed + 2 * (E - 1)
        params = T.encoder_params(model)
        cs, cc, ch = (SimpleNamespace(param_list=params) for _ in range(3))
        seg = SegLayout(E, S, Q, N)
        cs.seg = seg
        feat = T.EncoderTrainFn.forward(cs, batch.x_all.view(E * (S + Q), model.in_channels, N), model, seed)
        sfeat, qfeat = feat, feat[S * N:]
        cc.E = ch.E = E
        cc.ep_rows = ch.ep_rows = seg.ep_rows
        closs = contrast.ContrastFn.forward(cc, sfeat, model.proj.weight, model.proj.bias, model, batch.support_y,
                                            batch.support_flag)
        lploss = HeadLPFn.forward(ch, sfeat, qfeat, model, batch.support_y, batch.query_y)
        logits = model._train_logits
        metrics = contrast.train_debug_metrics(model, batch.support_y, batch.gt_support_y, batch.query_y, batch.gt_query_y,
                                               None, E=E)
        loss = lploss + loss_weight * closs
        # ---- backward, in dependency order
        one = torch.ones((), device=feat.device)
        dfeat_c, dWp, dbp = contrast.ContrastFn.backward(cc, one * loss_weight)[:3]
        HeadLPFn.backward(ch, one)
        dfeat = ch.dfeat_full  # per episode (support rows | query rows), the layout of `feat`
        assert dfeat.shape[0] == feat.shape[0]
        dfeat.add_(dfeat_c)  # (the contrast gradient has the batch's layout too: zero on the query rows)
        grads = T.EncoderTrainFn.backward(cs, dfeat)[3:]
        index = {id(p): i for i, p in enumerate(q for q in model.parameters() if q.requires_grad)}
        dst, src = [], []
        for p, g in zip(params, grads):
            if g is not None:
                dst.append(grad_sink[index[id(p)]])
                src.append(g.reshape(p.shape))
        dst += [grad_sink[index[id(model.proj.weight)]], grad_sink[index[id(model.proj.bias)]]]
        src += [dWp, dbp]
        torch._foreach_add_(dst, src)
    n_q = batch.query_x.shape[1]
    return (loss.reshape(E), logits.reshape(E, n_q, model.n_classes, N), metrics, lploss.reshape(E), closs.reshape(E))


def explicit_train_episode(model, episode, grad_sink, loss_weight=0.1):
    """One episode (train layout, loader.py:1666-1671) through explicit_train_batch: the launch sequence that
    episode_graph.EpisodeGraphs freezes into a hipGraph.  Returns (loss, logits, metrics[4], lp_loss, contrast_loss)."""
    from .batch import EpisodeBatch
    model._lp_force = False  # a frozen launch sequence always runs on the slot's fixed budget
    b = EpisodeBatch.from_episodes([episode])
    loss, logits, metrics, lp, con = explicit_train_batch(model, b, grad_sink, loss_weight)
    return loss.reshape(()), logits[0], tuple(metrics.reshape(4).unbind(0)), lp.reshape(()), con.reshape(())
