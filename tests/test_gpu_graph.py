"""Episodes replayed as captured hipGraphs on several HIP streams (r3dfsseg_amd/episode_graph.py) must give the
results of the eager launch sequence: logits for evaluation, the summed gradient for training."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _model(cfg, train):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(S.make_state_dict(cfg, 123))
    m.cuda().train(train)
    m.att_learner.dropout.p = 0.0  # eager and graph mode advance their dropout seeds differently
    return m


def _episodes(cfg, n):
    out = []
    for e in range(n):
        data, _ = S.make_episode(cfg, seed=40 + e, noise_ratio=0.2, train=True)
        out.append([t.cuda() for t in data])
    return out


def test_eval_graphs_match_eager():
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m = _model(cfg, False)
    eps = _episodes(cfg, 5)
    eager = []
    with torch.no_grad():
        for ep in eps:
            logits, _ = m(*ep[:4], lp_iters=m.lp_max_iter)
            assert m.lp_converged()
            eager.append(logits.clone())
    g = EpisodeGraphs(m, eps[0][:4], n_slots=2, train=False, lp_budget=120)
    out = torch.empty(len(eps), *eager[0].shape, device="cuda")
    for _ in range(2):  # the second pass replays graphs whose buffers hold another episode's leftovers
        loss_sum = g.run([ep[:4] for ep in eps], logits_out=out)
    torch.cuda.synchronize()
    bad, iters, mx = g.check()
    assert bad == 0 and 0 < mx <= 120
    for e in range(len(eps)):
        np.testing.assert_allclose(out[e].cpu().numpy(), eager[e].cpu().numpy(), atol=2e-5, rtol=1e-5)
        assert torch.equal(out[e].argmax(1), eager[e].argmax(1))
    assert np.isfinite(float(loss_sum))
    # the model's own eager path still works after capture (slot state restored)
    with torch.no_grad():
        logits, _ = m(*eps[1][:4], lp_iters=m.lp_max_iter)
    np.testing.assert_allclose(logits.cpu().numpy(), eager[1].cpu().numpy(), atol=2e-5, rtol=1e-5)
    # a weight update reaches the graphs: the folded weights they point at are refreshed in place
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.03)
        want, _ = m(*eps[2][:4], lp_iters=m.lp_max_iter)
    assert (want - eager[2]).abs().max().item() > 1e-3
    g.run([eps[2][:4]], logits_out=out[:1])
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[0].cpu().numpy(), want.cpu().numpy(), atol=2e-5, rtol=1e-5)


def test_cg_budget_by_disabling_graph_nodes():
    """The captured CG loop can be shortened without re-capturing (r3d_graph_set_lp_budget): same logits while
    the enabled iterations cover the solve, a reported miss when they do not, and full recovery afterwards."""
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m = _model(cfg, False)
    eps = [ep[:4] for ep in _episodes(cfg, 4)]
    g = EpisodeGraphs(m, eps[0], n_slots=2, train=False, lp_budget=120)
    g.adaptive_budget = False
    ref = torch.empty(len(eps), 2, 3, 512, device="cuda")
    g.run(eps, logits_out=ref)
    torch.cuda.synchronize()
    bad, _, mx = g.check()
    assert bad == 0 and 0 < mx < 100
    out = torch.empty_like(ref)
    g.set_lp_budget(mx + 2)          # convergence is detected by the launch after the last productive one
    assert g.active_budget == mx + 2
    g.run(eps, logits_out=out)
    torch.cuda.synchronize()
    assert g.check()[0] == 0
    assert torch.equal(out, ref)
    g.set_lp_budget(3)               # far too few: every replay must say so
    g.run(eps, logits_out=out)
    torch.cuda.synchronize()
    bad, _, _ = g.check()
    assert bad == len(eps) and g.last_unconverged == len(eps)
    assert g.active_budget == 120    # check() restores everything that was captured
    g.run(eps, logits_out=out)
    torch.cuda.synchronize()
    assert g.check()[0] == 0 and torch.equal(out, ref)
    # adaptive mode: the lagged probe shrinks the budget to 1.5 * max + 8 (rounded up to 8, at least 24)
    g.adaptive_budget = True
    for _ in range(4):
        g.run(eps, logits_out=out)
    torch.cuda.synchronize()
    assert g.active_budget == g.budget_for(mx) and g.active_budget < 120
    assert g.check()[0] == 0 and torch.equal(out, ref)


def test_train_graphs_accumulate_the_eager_gradient():
    from r3dfsseg_amd.dist import FlatGradBucket
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m = _model(cfg, True)
    eps = _episodes(cfg, 3)
    bucket = FlatGradBucket(m.parameters())
    running0 = m.encoder.conv.layer[1].running_mean.clone()
    losses = []
    for ep in eps:
        out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10])
        loss = out[1] + 0.1 * out[2]
        loss.backward()
        losses.append(float(loss))
    want = bucket.flat.clone()
    # restore the BatchNorm buffers: the graph run below starts from the same state
    m.load_state_dict(S.make_state_dict(cfg, 123))
    rows = torch.zeros(2, bucket.flat.numel(), device="cuda")
    g = EpisodeGraphs(m, eps[0], n_slots=2, train=True, lp_budget=150, grad_rows=rows)
    assert torch.equal(m.encoder.conv.layer[1].running_mean, running0)  # capture warm-up left no trace
    total = g.run(eps)
    torch.cuda.synchronize()
    bad, iters, mx = g.check()
    assert bad == 0
    got = rows.sum(0)
    assert abs(float(total) - sum(losses)) < 1e-4 * max(1.0, abs(sum(losses)))
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-5, err  # same kernels, same per-episode results: only the order of the sum over episodes differs
    # a second step reuses the graphs: rows are zeroed and re-accumulated -- bit-identical now that no kernel of the
    # step sums in a run-dependent order (batch statistics are per episode, so the weights being equal is all it needs)
    first = rows.sum(0).clone()
    g.run(eps)
    torch.cuda.synchronize()
    assert torch.equal(rows.sum(0), first)


def test_running_statistics_of_a_graph_step_equal_the_sequential_schedule():
    """Replays on several streams record their BatchNorm batch statistics; applied in episode order after the step
    they give bit for bit the running statistics of the reference's schedule (one episode after the other, two
    getFeatures calls each: mpti.py:434,436 -> 2 updates per BatchNorm and episode)."""
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m = _model(cfg, True)
    eps = _episodes(cfg, 5)
    for ep in eps:  # eager, sequential
        m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10])
    want = {k: v.clone() for k, v in m.named_buffers()}
    m.load_state_dict(S.make_state_dict(cfg, 123))
    g = EpisodeGraphs(m, eps[0], n_slots=2, train=True, lp_budget=150)
    g.run(eps)
    torch.cuda.synchronize()
    assert g.check()[0] == 0
    for k, v in m.named_buffers():
        assert torch.equal(v, want[k]), k
    assert int(m.encoder.conv.layer[1].num_batches_tracked) == 2 * len(eps)


def test_learner_graph_path_equals_eager_path():
    """MPTILearner_V3.train / .test with episode_graphs (one captured hipGraph per call) against the same learner with
    eager launches: same return tuples, same weights after three optimiser steps, same predictions."""
    from r3dfsseg_amd.mpti_learner import MPTILearner_V3
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512, pretrain_checkpoint_path="synthetic", model_checkpoint_path=None,
                     lr=1e-3, step_size=5000, gamma=0.5)
    eps = _episodes(cfg, 3)
    outs = {}
    for mode in (False, True):
        L = MPTILearner_V3(SimpleNamespace(**dict(cfg, episode_graphs=mode)), mode="train")
        L.model.att_learner.dropout.p = 0.0
        tuples = []
        for ep in eps:
            t = L.train(ep, None)
            assert len(t) == 8
            tuples.append([float(v) for v in t])
        pred, loss, acc = L.test(eps[0][:7], [3, 6], eval=True)
        pred2, loss2, acc2 = L.test(eps[1][:7], [3, 6], eval=False)
        outs[mode] = (tuples, torch.cat([p.detach().reshape(-1) for p in L.model.parameters()]).cpu(),
                      {k: v.cpu().clone() for k, v in L.model.named_buffers()}, pred.cpu(), float(loss), acc, pred2.cpu(), acc2)
        assert (L._trainer is not None) == mode and (len(L._eval_graphs) == 2) == mode
    te, tg = outs[False][0], outs[True][0]
    for a, b in zip(te, tg):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5)
    perr = (outs[False][1] - outs[True][1]).abs().max().item()
    assert perr < 2e-5, perr
    for k in outs[False][2]:
        np.testing.assert_allclose(outs[False][2][k].float().numpy(), outs[True][2][k].float().numpy(), rtol=1e-4, atol=1e-5)
    assert torch.equal(outs[False][3], outs[True][3]) and torch.equal(outs[False][6], outs[True][6])
    assert abs(outs[False][4] - outs[True][4]) < 1e-4 and outs[False][5] == outs[True][5] and outs[False][7] == outs[True][7]


def test_graph_weight_guard_of_the_multi_stream_schedule():
    """DPTrainer's runtime guard for schedules with several streams in flight (EpisodeGraphs.verify_graph_weights): the
    label-propagation edge weights every slot's last solve used, recomputed alone on the drained chip, must have the same
    bits -- and a single flipped bit in a slot's weights must be reported."""
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m = _model(cfg, False)
    eps = _episodes(cfg, 6)
    g = EpisodeGraphs(m, eps[0][:4], n_slots=3, train=False)
    g.run([e[:4] for e in eps])
    assert g.step_status()[:2] == (0, 0)
    assert g.verify_graph_weights() == 0
    hb = g.slots[1].state.last[1]
    nnz_cap = 2 * hb.n_cap * (hb.kp1 - 1)
    wdir0 = hb.lp_off["val"] + nnz_cap          # (the directed weights follow the symmetric values in the workspace)
    ws = hb.lp_ws.view(-1)
    ws[wdir0 + 5] ^= 1                           # one bit of one weight
    assert g.verify_graph_weights() == 1
    ws[wdir0 + 5] ^= 1
    assert g.verify_graph_weights() == 0
