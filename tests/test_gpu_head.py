"""GPU parity tests for the transductive head through the C ABI."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import r3d_oracle as O
from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from r3dfsseg_amd import ops as _ops
    from r3dfsseg_amd import _lib
    _lib.load()
    return _ops


def _segments(support_y):
    """Per segment (bg, fg way 0, ...): global point ids in the reference's nonzero order."""
    n_way, k_shot, N = support_y.shape
    flat = support_y.reshape(-1)
    segs = [torch.nonzero(flat == 0).squeeze(1)]
    for w in range(n_way):
        ids = torch.nonzero(support_y[w].reshape(-1) == 1).squeeze(1) + w * k_shot * N
        segs.append(ids)
    return segs


@pytest.mark.parametrize("n_way,k_shot,N,k_sub,seed", [(2, 2, 512, 100, 1), (3, 1, 256, 40, 2), (1, 1, 1024, 100, 3),
                                                       (5, 1, 256, 40, 4)])
def test_prototypes_bitexact_indices(ops, n_way, k_shot, N, k_sub, seed):
    cfg = S.make_cfg(n_way=n_way, k_shot=k_shot, pc_npts=N, n_subprototypes=k_sub)
    data, _ = S.make_episode(cfg, seed)
    support_y = data[1]
    Sx = n_way * k_shot
    D = 192
    rs = np.random.RandomState(seed + 100)
    feat = torch.from_numpy((rs.randn(Sx * N, D) * 0.1).astype(np.float32))
    # a few exact duplicate feature rows: FPS / argmin ties
    dup = rs.randint(0, Sx * N, 40)
    feat[dup[:20]] = feat[dup[20:]]
    n_q = n_way
    qfeat = torch.from_numpy((rs.randn(n_q * N, D) * 0.1).astype(np.float32))
    hb = ops.HeadBuffers(n_way, k_shot, N, n_q * N, k_sub, 200, D, "cuda")
    sfeat = feat.cuda()
    sy = support_y.reshape(Sx, N).to(torch.int32).contiguous().cuda()
    ops.head_prototypes(hb, sy, None, sfeat, qfeat.cuda())
    torch.cuda.synchronize()
    desc = hb.desc.cpu().numpy()
    ws = hb.proto_ws.cpu().numpy()
    comp_off, _, _, _, sel_off, seeds_off = hb.ws_off
    assign = hb.assign.cpu().numpy()
    nodes = hb.nodes.cpu()
    Y = torch.cat([hb.Y[p * hb.n_cap:(p + 1) * hb.n_cap] for p in range(hb.planes)], 1).cpu()  # (n_cap, 4 or 8) one-hot
    segs = _segments(support_y)
    Sn = Sx * N
    seg_off = [0] + [Sn + w * k_shot * N for w in range(n_way)]
    row = 0
    for s, ids in enumerate(segs):
        cnt = len(ids)
        assert desc[ops.HD_SEG_COUNT + s] == cnt
        comp = ws[comp_off + seg_off[s]: comp_off + seg_off[s] + cnt]
        assert np.array_equal(comp, ids.numpy()), "compaction order"
        f = feat[ids]
        if cnt > k_sub:
            kseg = O.fps_sample_count(cnt, k_sub)  # k or k + 1 (torch_cluster's float-rounded count)
            want_sel = O.fps(f, kseg).numpy()
            got_sel = ws[sel_off + s * 128: sel_off + s * 128 + kseg]
            assert np.array_equal(got_sel, want_sel), "FPS order differs in segment %d" % s
        protos, asg, m, seeds = O.get_multiple_prototypes(f, k_sub)
        assert desc[ops.HD_SEG_M + s] == m
        assert desc[ops.HD_SEG_POFF + s] == row
        assert np.array_equal(assign[seg_off[s]: seg_off[s] + cnt], asg.numpy()), "assignment differs in segment %d" % s
        np.testing.assert_allclose(nodes[row:row + m].numpy(), protos.numpy(), atol=1e-6, rtol=1e-5)
        lab = torch.zeros(m, 4 * hb.planes)
        lab[:, s] = 1
        assert torch.equal(Y[row:row + m], lab)
        row += m
    assert desc[ops.HD_N_PROTO] == row and desc[ops.HD_N_NODES] == row + n_q * N
    assert torch.equal(nodes[row:row + n_q * N], qfeat)
    assert (Y[row:row + n_q * N] == 0).all()


@pytest.mark.parametrize("fg_counts", [(149, 163), (364, 101), (150, 297)])
def test_prototype_count_follows_torch_cluster(ops, fg_counts):
    """torch_cluster.fps draws ceil(float32(n) * float32(k / n)) samples: 101 for n = 149, 163, 297, 364 at k = 100
    (models/mpti.py:612-613; tests/golden/head_eval.npz holds the n = 364 case as the reference's forward produced it).
    Segment sizes are forced through the masks; seeds, assignments and prototypes against the oracle."""
    n_way, k_shot, N, k_sub, D = 2, 1, 512, 100, 192
    rs = np.random.RandomState(sum(fg_counts))
    support_y = torch.zeros(n_way, k_shot, N, dtype=torch.int32)
    for w, c in enumerate(fg_counts):
        support_y[w, 0, torch.from_numpy(rs.permutation(N)[:c])] = 1
    feat = torch.from_numpy((rs.randn(n_way * N, D) * 0.1).astype(np.float32))
    qfeat = torch.from_numpy((rs.randn(n_way * N, D) * 0.1).astype(np.float32))
    hb = ops.HeadBuffers(n_way, k_shot, N, n_way * N, k_sub, 200, D, "cuda")
    for one_launch in (True, False):  # the persistent launch and the launch-per-round form
        hb.fps_one_launch = one_launch
        hb.desc.zero_()
        ops.head_prototypes(hb, support_y.reshape(n_way, N).contiguous().cuda(), None, feat.cuda(), qfeat.cuda())
        torch.cuda.synchronize()
        desc = hb.desc.cpu().numpy()
        nodes = hb.nodes.cpu()
        row = 0
        for s, ids in enumerate(_segments(support_y)):
            f = feat[ids]
            want = O.fps_sample_count(len(ids), k_sub)
            if s > 0:
                assert want == (101 if fg_counts[s - 1] in (149, 163, 297, 364) else 100)  # (n = 101 and 150 draw 100)
            protos, asg, m, _ = O.get_multiple_prototypes(f, k_sub)
            assert m == want and desc[ops.HD_SEG_M + s] == m, (s, m, want, desc[ops.HD_SEG_M + s])
            np.testing.assert_allclose(nodes[row:row + m].numpy(), protos.numpy(), atol=1e-6, rtol=1e-5)
            row += m
        assert desc[ops.HD_N_PROTO] == row


def _graph_nodes(n_proto, n_q_pts, seed, scale=0.06):
    rs = np.random.RandomState(seed)
    centers = rs.randn(3, 192).astype(np.float32) * scale * 2
    lab = rs.randint(0, 3, n_proto + n_q_pts)
    x = centers[lab] + rs.randn(n_proto + n_q_pts, 192).astype(np.float32) * scale
    Y = torch.zeros(n_proto + n_q_pts, 4)
    Y[torch.arange(n_proto), torch.from_numpy(lab[:n_proto])] = 1
    return torch.from_numpy(x), Y


@pytest.mark.parametrize("n_proto,n_q_pts,cap_extra", [(300, 1024, 0), (260, 1024, 40)])
def test_label_propagation_vs_closed_form(ops, n_proto, n_q_pts, cap_extra):
    """CG on the sparse graph == the reference's dense closed form inv(I - 0.99 S) Y (mpti.py:775)."""
    n = n_proto + n_q_pts
    x, Y = _graph_nodes(n_proto, n_q_pts, 5)
    hb = ops.HeadBuffers(2, 1, n_q_pts // 2, n_q_pts, (n_proto + cap_extra) // 3, 200, 192, "cuda")
    assert hb.n_cap == n + cap_extra + 3  # one spare prototype slot per class (k or k + 1 FPS samples)
    hb.nodes[:n] = x.cuda()
    hb.nodes[n:] = 7.0  # garbage beyond n must be ignored
    hb.Y.zero_()
    hb.Y[:n] = Y.cuda()
    hb.desc[ops.HD_N_PROTO] = n_proto
    hb.desc[ops.HD_N_NODES] = n
    nbr = ops.knn(hb.nodes, 1, hb.n_cap, 201, mode=ops.SCORE_L2, n_valid=hb.desc[ops.HD_N_NODES:], status=hb.knn_status)
    assert int(hb.knn_status.item()) == 0
    want_nbr = O.knn_l2(x, 201)
    assert np.array_equal(nbr.cpu().numpy()[0, :n].astype(np.int64), want_nbr.numpy()), "201-NN indices differ"
    Z = ops.label_propagate(hb, nbr, 1.0, 0.99, 300, 1e-6)
    torch.cuda.synchronize()
    conv, iters = hb.stats.cpu().tolist()
    assert conv == 1, "CG did not converge in 300 iterations"
    A = O.affinity(x, 200, 1.0)
    Zw = O.label_propagate(A, Y[:, :3])
    err = (Z[:n, :3].cpu() - Zw).abs().max().item()
    scale = Zw.abs().max().item()
    assert err <= TOL * max(1.0, scale), (err, scale, iters)
    # the residual of the reference's linear system, in float64
    Z64 = O.label_propagate(A, Y[:, :3], dtype=torch.float64)
    assert (Z[:n, :3].cpu().double() - Z64).abs().max().item() <= TOL * max(1.0, scale)
    print("CG iterations:", iters, "max|Z|:", scale, "err:", err)


@pytest.mark.parametrize("sep,n_proto", [(6.0, 300), (10.0, 300), (6.0, 20)])
def test_label_propagation_on_clustered_features(ops, sep, n_proto):
    """The trained regime: well separated feature clusters make the graph nearly disconnected, S gets one eigenvalue
    ~1 per cluster and plain CG needs 50-70 iterations (measured on systems dumped after 150 training steps).  The
    two-level solver must reach the same closed-form answer in a fraction of that, also when there are fewer
    prototypes than coarse seeds (empty aggregates) and when a right-hand side column is all zero."""
    n_q_pts = 2048
    n = n_proto + n_q_pts
    x, Y = _graph_nodes(n_proto, n_q_pts, 9, scale=0.06)
    rs = np.random.RandomState(10)
    lab = rs.randint(0, 3, n)
    x = x + torch.from_numpy(np.eye(3, 192, dtype=np.float32)[lab] * (sep * 0.06))  # push the clusters apart
    Y = torch.zeros(n, 4)
    Y[torch.arange(n_proto), torch.from_numpy(lab[:n_proto])] = 1
    hb = ops.HeadBuffers(2, 1, n_q_pts // 2, n_q_pts, (n_proto + 2) // 3, 200, 192, "cuda")
    assert hb.n_cap >= n
    hb.nodes[:n] = x.cuda()
    hb.Y.zero_()
    hb.Y[:n] = Y.cuda()
    hb.desc[ops.HD_N_PROTO] = n_proto
    hb.desc[ops.HD_N_NODES] = n
    nbr = ops.knn(hb.nodes, 1, hb.n_cap, 201, mode=ops.SCORE_L2, n_valid=hb.desc[ops.HD_N_NODES:], status=None)
    Z = ops.label_propagate(hb, nbr, 1.0, 0.99, 300, 1e-6)
    torch.cuda.synchronize()
    conv, iters = hb.stats.cpu().tolist()
    A = O.affinity(x, 200, 1.0)
    Z64 = O.label_propagate(A, Y[:, :3], dtype=torch.float64)
    scale = Z64.abs().max().item()
    err = (Z[:n, :3].cpu().double() - Z64).abs().max().item()
    # what plain CG would need on this system (float64, same tolerance)
    D = A.double().sum(1)
    Sm = A.double() / torch.sqrt(D[:, None] * D[None, :])
    M = torch.eye(n, dtype=torch.float64) - 0.99 * Sm
    b = Y[:, :3].double()
    xx, r = torch.zeros_like(b), b.clone()
    p, rr, plain = r.clone(), (r * r).sum(0), 0
    while (rr > 1e-12 * (b * b).sum(0)).any() and plain < 500:
        q = M @ p
        al = rr / (p * q).sum(0)
        xx += al * p; r -= al * q
        rn = (r * r).sum(0)
        p = r + (rn / rr) * p
        rr = rn; plain += 1
    print("sep %.0f n_proto %d: two-level CG %d iterations (plain CG %d), max|Z| %.1f err %.2e" % (sep, n_proto, iters, plain, scale, err))
    assert conv == 1 and err <= TOL * max(1.0, scale), (conv, iters, err, scale)
    assert (Z[:n, 3] == 0).all()                       # an all-zero right-hand side column stays zero
    if n_proto >= 64:
        assert iters <= max(12, plain // 2), (iters, plain)


def test_logits_and_cross_entropy(ops):
    n_q, N = 2, 512
    hb = ops.HeadBuffers(2, 1, N, n_q * N, 100, 200, 192, "cuda")
    n_proto = 287
    hb.desc[ops.HD_N_PROTO] = n_proto
    Z = torch.from_numpy(np.random.RandomState(9).randn(hb.n_cap, 4).astype(np.float32))
    hb.Z.copy_(Z)
    labels = torch.from_numpy(np.random.RandomState(10).randint(0, 3, (n_q, N)).astype(np.int64))
    logits, loss, pred = ops.query_logits_ce(hb, n_q, 3, labels.cuda())
    want_logits = Z[n_proto:n_proto + n_q * N, :3].view(n_q, N, 3).transpose(1, 2)
    assert torch.equal(logits.cpu(), want_logits.contiguous())
    want_loss = torch.nn.functional.cross_entropy(want_logits, labels)
    assert abs(loss.item() - want_loss.item()) < 1e-5
    assert torch.equal(pred.cpu().long(), want_logits.argmax(1))


@pytest.mark.parametrize("n_way,k_shot,N,seed", [(2, 1, 512, 1), (2, 2, 512, 2), (4, 1, 512, 3), (5, 2, 256, 4), (7, 1, 256, 5)])
def test_mpti_forward_eval_vs_oracle(ops, n_way, k_shot, N, seed):
    """Whole MPTI_SelfAtten.forward (eval) against the oracle restatement of mpti.py:414-577.  More than 3 ways (the
    reference takes any n_way, mpti.py:49,58): 5..8 label columns, solved as two planes of 4 on the same graph."""
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=n_way, k_shot=k_shot, pc_npts=N)
    sd = S.make_state_dict(cfg, 123)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().eval()
    data, _ = S.make_episode(cfg, seed)
    sx, sy, qx, qy = data[:4]
    m._trace = {}
    with torch.no_grad():
        logits, loss = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda(), lp_iters=m.lp_max_iter)
    assert m.lp_converged()
    # chain of custody (tests/custody.py): the head's index decisions on the HIP node matrix are the oracle's, and with
    # them every logit is within 1e-4 at every point
    from custody import head_custody
    head_custody(m, cfg, sd, data, logits, loss)
    # and end to end against the oracle on its OWN features and decisions: same predictions except behind near-ties
    (wl, wloss), aux = O.mpti_forward(sd, cfg, sx, sy, qx, qy, return_aux=True)
    agree = (logits.cpu().argmax(1) == wl.argmax(1)).float().mean().item()
    assert agree >= 0.99, agree
    assert abs(loss.item() - wloss.item()) <= 5e-3 * max(1.0, abs(wloss.item()))


def test_mpti_forward_without_attention_vs_oracle(ops):
    """use_attention = False (models/mpti.py:588-590: a bias-free 1x1 conv `linear_mapper` takes the place of the
    attention): the eval forward with the same chain of custody as above."""
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512, use_attention=False)
    sd = S.make_state_dict(cfg, 123)
    assert "linear_mapper.weight" in sd and not any(k.startswith("att_learner") for k in sd)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().eval()
    data, _ = S.make_episode(cfg, 4)
    sx, sy, qx, qy = data[:4]
    m._trace = {}
    with torch.no_grad():
        logits, loss = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda(), lp_iters=m.lp_max_iter)
    assert m.lp_converged()
    # the features themselves against the oracle with the HIP neighbour lists injected: 1e-4 on every point
    Sn, N = 4, 512
    for p, x in enumerate((sx.reshape(Sn, -1, N), qx)):
        b0 = 0 if p == 0 else Sn
        idx = [i[b0:b0 + x.shape[0]].cpu().to(torch.int64) for i in m._trace["idx"][0]]
        f = O.get_features(sd, x, cfg, idx_override=idx).transpose(1, 2).reshape(x.shape[0] * N, -1)
        got = (m._trace["sfeat"] if p == 0 else m._trace["qfeat"]).cpu()
        assert ((got - f).abs() / f.abs().clamp(min=1.0)).max().item() <= 1e-4
    from custody import head_custody
    head_custody(m, cfg, sd, data, logits, loss)


def test_device_sample_count_equals_the_host_arithmetic_for_every_n(ops):
    """hp_fps_count (double division, rounding to float32, float32 product, ceil) on the device against numpy's float32
    arithmetic -- the published torch_cluster count -- for every point count up to 65536 and several k."""
    from r3dfsseg_amd import _lib
    lib = _lib.load()
    n_max = 65537
    for k in (4, 40, 100, 127):
        out = torch.empty(n_max, device="cuda", dtype=torch.int32)
        _lib.check(lib.r3d_fps_sample_count_table(k, n_max, ops._p(out), ops._st()))
        got = out.cpu().numpy()
        n = np.arange(n_max)
        with np.errstate(divide="ignore", invalid="ignore"):
            want = np.ceil(n.astype(np.float32) * (k / n.astype(np.float64)).astype(np.float32)).astype(np.int64)
        want = np.where(n > k, np.minimum(want, n), n)
        assert np.array_equal(got, want), (k, np.nonzero(got != want)[0][:10])
        assert set(np.unique(got[k + 1:])) <= {k, k + 1}
    assert O.fps_sample_count(364, 100) == 101  # the oracle's statement of the same rule
