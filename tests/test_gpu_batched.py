"""Episode-batched execution (r3dfsseg_amd/batched.py): E episodes through ONE launch sequence must give, per episode,
the results of the one-episode path -- the reference's schedule (mpti_train_noise.py:72-98 runs one episode per step).
Forward results that depend on BatchNorm only are bit for bit equal (a segment's reductions are partitioned by the
segment, not by the batch); the attention output depends on the key-axis split, which follows the number of clouds in
the launch, so logits agree to rounding; weight gradients are summed over the batch in another order."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _model(cfg, train, p_drop=0.1):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(S.make_state_dict(cfg, 123))
    m.cuda().train(train)
    m.att_learner.dropout.p = p_drop
    m._lp_budget = 150  # the adaptive CG launch budget starts at 32: these tests are about equality, not about adaptation
    return m


def _episodes(cfg, n, noise=0.2):
    out = []
    for e in range(n):
        data, _ = S.make_episode(cfg, seed=70 + e, noise_ratio=noise, train=True)
        out.append([t.cuda() for t in data])
    return out


def _eager_train(cfg, eps, p_drop):
    from r3dfsseg_amd.dist import FlatGradBucket
    m = _model(cfg, True, p_drop)
    bucket = FlatGradBucket(m.parameters())
    per = []
    for ep in eps:
        m._trace = {}
        out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
                lp_iters=m.lp_max_iter)
        loss = out[1] + 0.1 * out[2]
        loss.backward()
        assert m.lp_converged(backward=True)
        per.append(dict(logits=out[0].detach().clone(), lp=out[1].item(), cl=out[2].item(),
                        metrics=[float(v) for v in out[3:]], sfeat=m._trace["sfeat"].detach().clone(),
                        qfeat=m._trace["qfeat"].detach().clone()))
    m._trace = None
    return m, bucket.flat.clone(), per, {k: v.clone() for k, v in m.named_buffers()}


@pytest.mark.parametrize("p_drop,n_way", [(0.0, 2), (0.1, 2), (0.1, 4)])  # (4 ways: two planes of label columns per system)
def test_batched_train_equals_one_episode_at_a_time(p_drop, n_way):
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.batched import EpisodeBatchRunner
    from r3dfsseg_amd.dist import FlatGradBucket
    cfg = S.make_cfg(n_way=n_way, k_shot=2, pc_npts=512)
    eps = _episodes(cfg, 3)
    _, want_grad, per, want_buf = _eager_train(cfg, eps, p_drop)
    m = _model(cfg, True, p_drop)
    bucket = FlatGradBucket(m.parameters())
    run = EpisodeBatchRunner(m)
    run.begin_step()
    m._trace = {}
    batch = EpisodeBatch.from_episodes(eps)
    loss, logits, metrics, lp, cl = run.train_batch(batch, [p.grad for p in bucket.params])
    bad, ovf, its, mx = run.step_status()
    assert bad == 0 and ovf == 0 and 0 < mx <= 150 and its >= 3
    Sn, N = 4, 512
    feat = None
    for e, w in enumerate(per):
        assert abs(lp[e].item() - w["lp"]) <= 2e-5 * max(1.0, abs(w["lp"])), (e, lp[e].item(), w["lp"])
        assert abs(cl[e].item() - w["cl"]) <= 2e-5 * max(1.0, abs(w["cl"])), (e, cl[e].item(), w["cl"])
        np.testing.assert_allclose(logits[e].cpu().numpy(), w["logits"].cpu().numpy(), atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(metrics[e].cpu().numpy(), np.array(w["metrics"], dtype=np.float32), atol=1e-6)
    # the statistics records were deferred: nothing touched the running statistics yet; applied, they are bit for bit
    # those of the one-episode-at-a-time schedule
    fresh = S.make_state_dict(cfg, 123)
    for k, v in m.named_buffers():
        assert torch.equal(v.cpu(), fresh[k]), k
    run.apply_running_stats()
    for k, v in m.named_buffers():
        assert torch.equal(v, want_buf[k]), k
    assert int(m.encoder.conv.layer[1].num_batches_tracked) == 2 * len(eps)
    err = (bucket.flat - want_grad).abs().max().item() / want_grad.abs().max().item()
    assert err < 1e-5, err
    # the batch again: same buffers, bit-identical gradient (nothing in the step sums in a run-dependent order)
    first = bucket.flat.clone()
    bucket.zero_()
    m._drop_seed = 0
    run.begin_step()
    run.train_batch(batch, [p.grad for p in bucket.params])
    assert run.step_status()[0] == 0
    assert torch.equal(bucket.flat, first)


def test_batched_features_are_bitwise_those_of_the_single_episode():
    """Everything in front of the head that BatchNorm (not the attention split) decides: level-1 and BaseLearner feature
    columns of every episode of a batch equal, bit for bit, the same episode run alone."""
    from r3dfsseg_amd import train_ops as T
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.ops import SegLayout
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    eps = _episodes(cfg, 3)
    Sn, Q, N = 4, 2, 512
    m = _model(cfg, True, 0.0)
    singles = []
    with torch.no_grad():
        T.update_running_stats = False
        try:
            for ep in eps:
                c = SimpleNamespace(param_list=T.encoder_params(m), seg=SegLayout(1, Sn, Q, N))
                x = torch.cat((ep[0].reshape(Sn, -1, N), ep[2]), 0)
                singles.append(T.EncoderTrainFn.forward(c, x, m, 0).clone())
            b = EpisodeBatch.from_episodes(eps)
            c = SimpleNamespace(param_list=T.encoder_params(m), seg=SegLayout(3, Sn, Q, N))
            feat = T.EncoderTrainFn.forward(c, b.x_all.view(3 * (Sn + Q), -1, N), m, 0)
        finally:
            T.update_running_stats = True
    rows = (Sn + Q) * N
    for e in range(3):
        got = feat[e * rows:(e + 1) * rows]
        assert torch.equal(got[:, :64], singles[e][:, :64]) and torch.equal(got[:, 128:], singles[e][:, 128:]), e
        assert (got[:, 64:128] - singles[e][:, 64:128]).abs().max().item() < 1e-5


@pytest.mark.parametrize("eval_flag,n_way", [(False, 2), (True, 2), (True, 5)])
def test_batched_eval_forward_equals_single_episodes(eval_flag, n_way):
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.batched import EpisodeBatchRunner
    cfg = S.make_cfg(n_way=n_way, k_shot=2, pc_npts=512)
    eps = _episodes(cfg, 5, noise=0.5)
    m = _model(cfg, False)
    want = []
    with torch.no_grad():
        for ep in eps:
            logits, loss = m(*ep[:4], eval=eval_flag, lp_iters=m.lp_max_iter)
            assert m.lp_converged()
            want.append((logits.clone(), loss.item()))
    run = EpisodeBatchRunner(m)
    run.begin_step()
    logits, loss = run.eval_batch(EpisodeBatch.from_episodes(eps), eval=eval_flag)
    assert run.step_status()[:2] == (0, 0)
    for e, (wl, wloss) in enumerate(want):
        np.testing.assert_allclose(logits[e].cpu().numpy(), wl.cpu().numpy(), atol=2e-5, rtol=1e-5)
        assert torch.equal(logits[e].argmax(1), wl.argmax(1))
        assert abs(loss[e].item() - wloss) < 2e-5 * max(1.0, abs(wloss))
    # several persistent FPS launches for one batch (the chip holds fps_slots workgroups of that kernel at once): same result
    hb = m._head[1]
    assert hb.E == 5 and hb.fps_group >= 5
    hb.fps_slots = 2 * hb.fps_blocks  # two episodes per launch: 2 + 2 + 1
    assert hb.fps_group == 2
    run.begin_step()
    logits2, _ = run.eval_batch(EpisodeBatch.from_episodes(eps), eval=eval_flag)
    assert run.step_status()[:2] == (0, 0) and torch.equal(logits2, logits)
    # ... and one launch per FPS round
    m._slot.fps_one_launch = False
    run.begin_step()
    logits3, _ = run.eval_batch(EpisodeBatch.from_episodes(eps), eval=eval_flag)
    assert run.step_status()[:2] == (0, 0) and torch.equal(logits3, logits)
    m._slot.fps_one_launch = True


def test_dptrainer_batched_step_equals_eager_step():
    """DPTrainer(batch_size=2) on 3 episodes (batches of 2 + 1) against the eager trainer on the same episodes: same
    weights after Adam, same running statistics; and an under-budgeted step is redone exactly (fail closed)."""
    from r3dfsseg_amd.dp_train import DPTrainer
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    eps = _episodes(cfg, 3)
    res = {}
    for mode in ("eager", "batched"):
        m = _model(cfg, True, 0.1)
        learner = SimpleNamespace(model=m)
        learner.optimizer = torch.optim.Adam(
            [{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
             {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
        learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
        tr = DPTrainer(learner, batch_size=2 if mode == "batched" else 0)
        l1 = float(tr.step(eps))
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        after1 = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        l2 = float(tr.step(eps))
        assert tr.n_redone == 0
        res[mode] = (l1, l2, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone(),
                     {k: v.clone() for k, v in m.named_buffers()}, grads, after1)
        names = [(n, p.detach()) for n, p in m.named_parameters()]
        if mode == "batched":
            assert tr.last_status[0] == 0 and tr.last_status[3] > 0
            m._lp_budget = 2  # far too few CG launches: the step must notice and redo its episodes on the exact schedule
            m._lp_probe = None
            l3 = float(tr.step(eps))
            assert tr.redone and tr.n_redone == 1 and np.isfinite(l3)
    a, b = res["eager"], res["batched"]
    # first step: the same weights, losses equal to rounding; second step: the weights have taken one Adam step, which
    # turns rounding-level differences of noise-sized gradient elements into lr-sized weight differences (below)
    assert abs(a[0] - b[0]) < 2e-5 * max(1.0, abs(a[0])) and abs(a[1] - b[1]) < 2e-3 * max(1.0, abs(a[1]))
    # the step's gradients agree to rounding (the batch adds the episodes' weight gradients in another association) ...
    for n, g in a[4].items():
        assert (g - b[4][n]).abs().max().item() <= 2e-6 * max(g.abs().max().item(), 1e-3), n
    # ... and so do the weights after the Adam step WHERE the gradient is not itself rounding noise: Adam normalises every
    # element to a step of ~lr (1e-3 here) whatever its size, so an element whose gradient is below the noise moves by
    # +-lr in either run (2 lr apart at worst).  Elements with |g| >= 1e-3 of their tensor's largest must agree to 1e-6.
    # (After a SECOND step the runs have drifted by more than rounding -- the noise-sized elements have moved the
    # weights by lr -- so only the first step is held to this; the second one's loss is compared above, its weights are
    # bounded by 4 lr.)
    perr1 = (a[5] - b[5]).abs()
    assert perr1.max().item() < 2.1e-3, perr1.max().item()
    masks = []
    for n, p in names:  # (the order of model.parameters(), i.e. of the concatenated weight vectors)
        g = a[4].get(n)
        masks.append((g.abs() >= 1e-3 * g.abs().max()).reshape(-1) if g is not None
                     else torch.zeros(p.numel(), dtype=torch.bool, device=p.device))
    solid = torch.cat(masks)
    assert solid.float().mean().item() > 0.5
    assert perr1[solid].max().item() < 1e-6, perr1[solid].max().item()
    assert (a[2] - b[2]).abs().max().item() < 4.1e-3
    for k in a[3]:  # (running statistics after the second step: within the drift described above)
        np.testing.assert_allclose(a[3][k].float().cpu().numpy(), b[3][k].float().cpu().numpy(), rtol=1e-3, atol=2e-4)


def test_batch_graph_replay_equals_eager_batch():
    """DPTrainer(batch_size=E, batch_graph=True): the batch's launch sequence as ONE captured hipGraph (batched.BatchGraph)
    against the same trainer launching eagerly -- the same kernels with the same arguments, so with the dropout off (the
    two keep different seed counters) every gradient, weight and running statistic agrees bit for bit over several steps,
    new episodes are copied into the static inputs, and an under-budgeted solve is noticed and redone (fail closed)."""
    from r3dfsseg_amd.dp_train import DPTrainer
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    sets = [_episodes(cfg, 3), [[t.clone() for t in ep] for ep in _episodes(cfg, 6)[3:]]]
    res = {}
    for mode in ("eager", "graph"):
        m = _model(cfg, True, 0.0)
        learner = SimpleNamespace(model=m)
        learner.optimizer = torch.optim.Adam(
            [{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
             {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
        learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
        tr = DPTrainer(learner, batch_size=3, batch_graph=(mode == "graph"))
        losses, grads = [], None
        for it in range(4):
            losses.append(float(tr.step(sets[it & 1])))
            if it == 0:
                grads = tr.bucket.flat.clone()
        assert tr.n_redone == 0 and tr.last_status[0] == 0
        outs = [tuple(t.clone() for t in o) for o in tr.last_outputs]
        res[mode] = (losses, grads, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone(),
                     {k: v.clone() for k, v in m.named_buffers()}, outs)
        if mode == "graph":
            g = tr.runner._graph
            assert g is not None and g.E == 3
            g.set_lp_budget(2)  # far too few CG iterations enabled: the step must notice and redo its episodes exactly
            l = float(tr.step(sets[0]))
            assert tr.redone and tr.n_redone == 1 and np.isfinite(l)
            assert g.active_budget == g.lp_budget  # a miss returns the budget to everything that was captured
    a, b = res["eager"], res["graph"]
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    for oa, ob in zip(a[4], b[4]):
        for ta, tb in zip(oa, ob):
            assert torch.equal(ta, tb)


def test_lds_resident_spmv_gives_the_same_bits():
    """The CG's SpMV has two forms (one matrix row per wave gathering r from L2; 128-row workgroups with r staged in LDS,
    chosen when a launch holds enough systems): same arithmetic, same <p, q> partials -> bit-identical solves, forward
    and adjoint."""
    from r3dfsseg_amd import _lib
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.batched import EpisodeBatchRunner
    from r3dfsseg_amd.dist import FlatGradBucket
    lib = _lib.load()
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    eps = _episodes(cfg, 3)
    batch = EpisodeBatch.from_episodes(eps)
    res = {}
    old = lib.r3d_debug_set_cg_spmv_lds_min_blocks(256)
    try:
        for mode, min_blocks in (("global", 1 << 30), ("lds", 0)):
            lib.r3d_debug_set_cg_spmv_lds_min_blocks(min_blocks)
            m = _model(cfg, True, 0.0)
            bucket = FlatGradBucket(m.parameters())
            run = EpisodeBatchRunner(m)
            run.begin_step()
            loss, logits, metrics, lp, cl = run.train_batch(batch, [p.grad for p in bucket.params])
            assert run.step_status()[:2] == (0, 0)
            hb = m._head[1]
            res[mode] = (logits.clone(), hb.Z.clone(), bucket.flat.clone(), hb.stats.clone(), hb.stats_bwd.clone())
    finally:
        lib.r3d_debug_set_cg_spmv_lds_min_blocks(old)
    for a, b in zip(res["global"], res["lds"]):
        assert torch.equal(a, b)
