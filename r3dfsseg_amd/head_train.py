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
        if model._lp_force:  # the conservative re-run (see MPTILearner_V3.train): one FPS launch per round as well
            hb.fps_one_launch = False
        ops.head_prototypes(hb, sy, None, sfeat, sfeatT, qfeat)
        nbr = ops.knn(hb.nodes, 1, hb.n_cap, hb.kp1, mode=ops.SCORE_L2, n_valid=hb.desc[ops.HD_N_NODES:],
                      status=None if model._lp_force else hb.knn_status)
        if model._lp_force:
            hb.knn_status.zero_()
        if model._trace is not None:
            model._trace["nbr"] = nbr
        # same launch-budget policy as eval (mpti.py: _lp_next_budget); MPTILearner_V3.train / DPTrainer.step check
        # lp_converged(backward=True) before the optimiser step and redo the episode on this conservative schedule
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
                                               int(min(model.lp_max_iter, ctx.budget + max(4, ctx.budget // 4))), float(model.lp_tol), _p(lam),
                                               _p(dnodes), D,
                                               _p(hb.lp_ws), hb.lp_ws.numel(), _p(hb.stats_bwd), _st()))
        # one buffer, support rows then query rows: the encoder backward takes it whole when both passes share launches
        dfeat = torch.empty(ctx.shapes[0][0] + ctx.shapes[1][0], D, device=dev, dtype=torch.float32)
        dsfeat, dqfeat = dfeat[:ctx.shapes[0][0]], dfeat[ctx.shapes[0][0]:]
        dsfeat.zero_()
        _lib.check(lib.r3d_head_prototypes_bwd(_p(dnodes), D, model.n_way, model.k_shot, N, D, n_q * N, _p(hb.desc),
                                               _p(hb.assign), _p(hb.cluster_count), _p(hb.proto_ws), _p(dsfeat), D,
                                               _p(dqfeat), D, _st()))
        return dsfeat, dqfeat, None, None, None


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
    # two getFeatures calls, each with its own BatchNorm batch statistics (mpti.py:434,436)
    if T.shared_launches_ok(model, S):  # ... through one launch sequence over the S + Q clouds
        sfeat, qfeat = T.get_features_train(model, torch.cat((sx, query_x), 0), seed, seg_clouds=[S, query_x.shape[0]])
    else:
        sfeat = T.get_features_train(model, sx, seed)
        qfeat = T.get_features_train(model, query_x, seed + 1)
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


def explicit_train_episode(model, episode, grad_sink, loss_weight=0.1):
    """One episode's forward + backward as a FIXED launch sequence without the autograd engine (the form that
    episode_graph.EpisodeGraphs freezes into a hipGraph): the forward halves of the three autograd Functions run
    with plain namespaces as their ctx, then their backward halves in dependency order, and every parameter
    gradient is ADDED into grad_sink[i] (views in the order of model.parameters(), requires_grad only).
    Same kernels, same results as ``loss = lp + loss_weight * contrast; loss.backward()``
    (models/mpti_learner.py:66-68).  Returns (loss, logits, metrics[4], lp_loss, contrast_loss)."""
    from types import SimpleNamespace
    from . import contrast
    (support_x, support_y, query_x, query_y, _sc, _qc, gt_support_y, gt_query_y, _bx, _by, support_flag) = episode
    S, N = model.n_way * model.k_shot, model.n_points
    with torch.no_grad():
        model._lp_force = False  # a frozen launch sequence always runs on the slot's fixed budget
        slot = model._slot
        if slot.seed_dev is not None:
            slot.seed_dev.add_(2)
            seed = 0
        else:
            model._drop_seed = getattr(model, "_drop_seed", 0) + 2
            seed = model._drop_seed
        params = T.encoder_params(model)
        cs, cq, cc, ch = (SimpleNamespace(param_list=params) for _ in range(4))
        sx = support_x.reshape(S, model.in_channels, N)
        shared = T.shared_launches_ok(model, S)
        if shared:  # both getFeatures calls through one launch sequence (BatchNorm statistics stay per call)
            cs.seg_clouds = [S, query_x.shape[0]]
            feat = T.EncoderTrainFn.forward(cs, torch.cat((sx, query_x), 0), model, seed)
            sfeat, qfeat = feat[:S * N], feat[S * N:]
        else:
            if T.bn_recorder is not None:
                T.bn_recorder.pass_id = 0
            sfeat = T.EncoderTrainFn.forward(cs, sx, model, seed)
            if T.bn_recorder is not None:
                T.bn_recorder.pass_id = 1
            qfeat = T.EncoderTrainFn.forward(cq, query_x, model, seed + 1)
        closs = contrast.ContrastFn.forward(cc, sfeat, model.proj.weight, model.proj.bias, model, support_y, support_flag)
        lploss = HeadLPFn.forward(ch, sfeat, qfeat, model, support_y, query_y)
        logits = model._train_logits
        metrics = contrast.train_debug_metrics(model, support_y, gt_support_y, query_y, gt_query_y, None)
        loss = lploss + loss_weight * closs
        # ---- backward, in dependency order
        one = torch.ones((), device=sfeat.device)
        dsf_c, dWp, dbp = contrast.ContrastFn.backward(cc, one * loss_weight)[:3]
        dsf, dqf = HeadLPFn.backward(ch, one)[:2]
        dsf.add_(dsf_c)
        if shared:  # dsf | dqf are the two halves of one buffer (HeadLPFn.backward)
            dfeat = dsf._base
            assert dfeat is not None and dfeat is dqf._base and dfeat.shape[0] == feat.shape[0]
            passes = (T.EncoderTrainFn.backward(cs, dfeat)[3:],)
        else:
            passes = (T.EncoderTrainFn.backward(cs, dsf)[3:], T.EncoderTrainFn.backward(cq, dqf)[3:])
        index = {id(p): i for i, p in enumerate(q for q in model.parameters() if q.requires_grad)}
        # one multi-tensor add per pass: a destination must not appear twice inside one foreach launch
        for k, grads in enumerate(passes):
            dst, src = [], []
            for p, g in zip(params, grads):
                if g is not None:
                    dst.append(grad_sink[index[id(p)]])
                    src.append(g.reshape(p.shape))
            if k == 0:
                dst += [grad_sink[index[id(model.proj.weight)]], grad_sink[index[id(model.proj.bias)]]]
                src += [dWp, dbp]
            torch._foreach_add_(dst, src)
    return loss, logits, metrics, lploss, closs
