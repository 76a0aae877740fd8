"""GPU parity of the TRAINING path: forward with batch-statistics BatchNorm and every gradient,
against torch-CPU autograd through the oracle restatement (same neighbour lists injected)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import r3d_oracle as O
from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(1e-12, b.abs().max().item())


def test_conv_bn_layer_fwd_bwd():
    from r3dfsseg_amd import ops, train_ops as T
    rs = np.random.RandomState(0)
    M, K, C = 3000, 192, 512
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32))
    W = torch.from_numpy((rs.randn(C, K) / np.sqrt(K)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(C)
    bn.weight.data = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32))
    bn.bias.data = torch.from_numpy(rs.uniform(-0.2, 0.2, C).astype(np.float32))
    R = torch.from_numpy(rs.randn(M, C).astype(np.float32))
    # reference: torch autograd on CPU
    xr, Wr = x.clone().requires_grad_(), W.clone().requires_grad_()
    bnr = torch.nn.BatchNorm1d(C)
    bnr.load_state_dict(bn.state_dict())
    y = torch.nn.functional.leaky_relu(bnr(xr @ Wr.t()), 0.2)
    (y * R).sum().backward()
    bng = bn.cuda()
    yg, saved = T.conv_bn_fwd(x.cuda(), W.cuda(), bng, ops.ACT_LRELU)
    np.testing.assert_allclose(yg.cpu().numpy(), y.detach().numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(bng.running_mean.cpu().numpy(), bnr.running_mean.numpy(), atol=1e-6)
    np.testing.assert_allclose(bng.running_var.cpu().numpy(), bnr.running_var.numpy(), atol=1e-5, rtol=1e-5)
    dW, dg, db, dbias, dX = T.conv_bn_bwd(saved, R.cuda())
    assert _rel(dW.cpu(), Wr.grad) < 2e-4
    assert _rel(dX.cpu(), xr.grad) < 2e-4
    assert _rel(dg.cpu(), bnr.weight.grad) < 2e-4 and _rel(db.cpu(), bnr.bias.grad) < 2e-4
    assert dbias.abs().max().item() < 1e-2  # a bias in front of batch-stat BN has zero gradient


def test_conv_bn_layer_is_exact_but_for_kink_flips():
    """conv + batch-statistics BatchNorm + LeakyReLU, forward and backward, against torch in float64 at the row count of a
    full-size episode, in BOTH matrix arithmetics (fp32 core; bf16 core with three-piece operands, csrc/gemm_bx3.hip).
    A gradient of this layer sums 10^7 terms behind the LeakyReLU kink: an activation within rounding of 0 takes slope 1
    in one implementation and 0.2 in the other, and that single element moves dW / dX by up to a few 1e-2 of their largest
    entry.  The flips are counted here (sign of the output against float64), which separates the two effects: with no
    flip every result is within 5e-6 of float64, in either arithmetic; a case with a flip only has to agree on y."""
    from r3dfsseg_amd import _lib, ops, train_ops as T
    lib = _lib.load()
    before = lib.r3d_get_matrix_arith()
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()
    clean = {0: 0, 1: 0}
    try:
        for seed, (M, K, C) in enumerate([(24576, 192, 512), (24576, 512, 256), (24576, 64, 128), (24576, 256, 128), (12288, 128, 64)]):
            rs = np.random.RandomState(seed)
            x = torch.from_numpy(rs.randn(M, K).astype(np.float32))
            x = torch.relu(x) * 1.3 + 0.05 * x  # (what a layer sees: the previous layer's activations)
            W = torch.from_numpy((rs.randn(C, K) / np.sqrt(K)).astype(np.float32))
            g = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32))
            b = torch.from_numpy(rs.uniform(-0.2, 0.2, C).astype(np.float32))
            R = torch.from_numpy(rs.randn(M, C).astype(np.float32))
            xr, Wr = x.double().requires_grad_(), W.double().requires_grad_()
            bnr = torch.nn.BatchNorm1d(C).double()
            bnr.weight.data, bnr.bias.data = g.double(), b.double()
            y = torch.nn.functional.leaky_relu(bnr(xr @ Wr.t()), 0.2)
            (y * R.double()).sum().backward()
            for arith in (0, 1):
                _lib.check(lib.r3d_set_matrix_arith(arith))
                bn = torch.nn.BatchNorm1d(C)
                bn.weight.data, bn.bias.data = g.clone(), b.clone()
                bn = bn.cuda()
                yg, saved = T.conv_bn_fwd(x.cuda(), W.cuda(), bn, ops.ACT_LRELU)
                dW, dg, db, _, dX = T.conv_bn_bwd(saved, R.cuda())
                assert rel(yg, y.detach()) < 5e-6
                assert rel(bn.running_mean, bnr.running_mean) < 1e-6 and rel(bn.running_var, bnr.running_var) < 1e-5
                flips = int(((yg.cpu() > 0) != (y.detach() > 0)).sum())
                if flips == 0:
                    clean[arith] += 1
                    errs = (rel(dW, Wr.grad), rel(dX, xr.grad), rel(dg, bnr.weight.grad), rel(db, bnr.bias.grad))
                    assert max(errs) < 5e-6, (arith, (M, K, C), errs)
    finally:
        _lib.check(lib.r3d_set_matrix_arith(before))
    assert clean[0] >= 2 and clean[1] >= 2, clean  # (each arithmetic proven on at least two flip-free layers)


def _encoder_setup(B, N, p_drop=0.0):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=1, pc_npts=N)
    sd = S.make_state_dict(cfg, 123, feat_scale=1.0)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().train()
    m.att_learner.dropout.p = p_drop
    pc = torch.from_numpy(np.stack([S._cloud(np.random.RandomState(90 + i), N, 0.0).T for i in range(B)]).copy())
    return cfg, sd, m, pc


def test_encoder_train_forward_and_gradients():
    """getFeatures in training mode against the oracle in float64 with the index decisions injected (neighbour lists AND
    max-pool winners: a winner that flips on an activation tie routes a whole gradient elsewhere and says nothing about
    the kernels).  Features 1e-4 on every point; every parameter gradient 2e-3 in the relative L2 norm (the full-size
    test holds 1e-3 over 24 576 points; here 1 024 points with a random upstream gradient leave BatchNorm's cancellations
    less to average over: measured 1.2e-3 at worst)."""
    from r3dfsseg_amd import ops, train_ops as T
    B, N = 2, 512
    cfg, sd, m, pc = _encoder_setup(B, N)
    R = torch.from_numpy(np.random.RandomState(5).randn(B * N, 192).astype(np.float32))
    m._trace = {}
    feat = T.get_features_train(m, pc.cuda(), seed=7)
    (feat * R.cuda()).sum().backward()
    idx = [i.cpu().to(torch.int64) for i in m._trace["idx"][0]]
    am = [a.cpu().to(torch.int64).view(B, N, 64).permute(0, 2, 1).contiguous() for a in m._trace["argmax"][0]]
    m._trace = None
    sdr = {k: (v.double().requires_grad_() if v.dtype.is_floating_point and "running" not in k
               else (v.double() if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    ns = {}
    fo = O.get_features(sdr, pc.double(), cfg, train=True, new_stats=ns, idx_override=idx, argmax_override=am)  # (B,192,N)
    fo_pm = fo.transpose(1, 2).reshape(B * N, 192)
    d = ((feat.detach().cpu().double() - fo_pm.detach()).abs() / fo_pm.detach().abs().clamp(min=1.0)).max().item()
    assert d <= 1e-4, d
    (fo_pm * R.double()).sum().backward()
    worst = []
    for name, p in m.named_parameters():
        if name.startswith("proj."):
            continue
        gref = sdr[name].grad
        assert p.grad is not None, name
        if name.startswith("base_learner") and name.endswith(".0.bias"):
            # a conv bias in front of batch-statistics BN has an exactly zero gradient (only rounding noise)
            assert p.grad.abs().max().item() < 1e-3 and gref.abs().max().item() < 1e-3
            continue
        g = p.grad.cpu().double()
        worst.append((((g - gref).norm() / gref.norm()).item(), _rel(g, gref), name))
    print("gradient errors (rel-L2, max-rel):", sorted(worst)[-4:])
    assert max(e for e, _, _ in worst) <= 2e-3, sorted(worst)[-4:]
    assert max(e for _, e, _ in worst) <= 1e-2, sorted(worst, key=lambda t: t[1])[-4:]  # (single LeakyReLU kink flips)
    for k, v in ns.items():  # running statistics follow nn.BatchNorm's update
        got = dict(m.named_buffers())[k].cpu()
        np.testing.assert_allclose(got.numpy(), v.float().numpy(), atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("K", [4, 8, 12, 16, 20, 24, 28, 32])
def test_edgeconv_train_layer_every_k(K):
    """One EdgeConv layer in training mode (forward + backward kernels, one instantiation per K) against torch
    autograd in fp64 on the same neighbour lists (models/dgcnn.py:26-61 with batch-statistics BatchNorm)."""
    from r3dfsseg_amd import train_ops as T
    B, N, C = 3, 72, 8
    rs = np.random.RandomState(100 + K)
    x = torch.from_numpy(rs.randn(B * N, C).astype(np.float32))
    idx = torch.from_numpy(np.stack([[rs.permutation(N)[:K] for _ in range(N)] for _ in range(B)]).astype(np.int32))
    conv1 = torch.nn.Conv2d(2 * C, 64, 1, bias=False)
    conv2 = torch.nn.Conv2d(64, 64, 1, bias=False)
    bn1, bn2 = torch.nn.BatchNorm2d(64), torch.nn.BatchNorm2d(64)
    with torch.no_grad():
        conv1.weight.copy_(torch.from_numpy(rs.randn(64, 2 * C, 1, 1).astype(np.float32) * 0.4))
        conv2.weight.copy_(torch.from_numpy(rs.randn(64, 64, 1, 1).astype(np.float32) * 0.2))
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.from_numpy(rs.uniform(-1.5, 1.5, 64).astype(np.float32)))  # both signs: max AND min winners
            bn.bias.copy_(torch.from_numpy(rs.randn(64).astype(np.float32) * 0.3))
    R = torch.from_numpy(rs.randn(B * N, 64).astype(np.float32))
    # reference, fp64
    xr = x.double().requires_grad_()
    W1r, W2r = conv1.weight.detach().double().reshape(64, 2 * C).requires_grad_(), conv2.weight.detach().double().reshape(64, 64).requires_grad_()
    g1, b1, g2, b2 = [t.detach().double().requires_grad_() for t in (bn1.weight, bn1.bias, bn2.weight, bn2.bias)]
    xb = xr.view(B, N, C)
    nb = torch.stack([xb[b][idx[b].long()] for b in range(B)])            # (B,N,K,C)
    ctr = xb[:, :, None, :].expand(B, N, K, C)
    e1 = torch.cat((nb - ctr, ctr), -1) @ W1r.t()                           # (B,N,K,64)

    def bn_act(z, g, b):
        mu, var = z.mean((0, 1, 2)), z.var((0, 1, 2), unbiased=False)
        return torch.nn.functional.leaky_relu((z - mu) / torch.sqrt(var + 1e-5) * g + b, 0.2)
    out_ref = bn_act(bn_act(e1, g1, b1) @ W2r.t(), g2, b2).max(2)[0].reshape(B * N, 64)
    (out_ref * R.double()).sum().backward()
    # device
    ec = SimpleNamespace(layer=[conv1.cuda(), bn1.cuda().train(), None, conv2.cuda(), bn2.cuda().train()])
    out = torch.empty(B * N, 64, device="cuda")
    saved = T.edgeconv_train_fwd(x.cuda(), idx.cuda().contiguous(), ec, B, N, out)
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.detach().numpy(), atol=2e-4, rtol=2e-4)
    dx = torch.zeros(B * N, C, device="cuda")
    dW1, dg1, db1, dW2, dg2, db2 = T.edgeconv_train_bwd(saved, R.cuda(), B, N, dx)
    torch.cuda.synchronize()
    for name, got, ref in (("dW1", dW1.reshape(64, 2 * C), W1r.grad), ("dW2", dW2.reshape(64, 64), W2r.grad), ("dg1", dg1, g1.grad),
                           ("db1", db1, b1.grad), ("dg2", dg2, g2.grad), ("db2", db2, b2.grad), ("dx", dx, xr.grad)):
        assert _rel(got.cpu().double(), ref) < 5e-4, (name, K, _rel(got.cpu().double(), ref))


@pytest.mark.parametrize("B,N,K,S,Q", [(4, 2048, 20, 3, 1), (6, 136, 12, 2, 1), (2, 512, 32, 2, 0)])
def test_edgeconv_backward_same_in_both_arithmetics(B, N, K, S, Q):
    """The EdgeConv backward on the bf16 matrix core in three-piece arithmetic (csrc/edgeconv_bwd_bx3.h: the default)
    against the fp32-core kernels (r3d_set_matrix_arith(0)) on the same forward: every output to 2e-5 of its largest entry
    (measured: ~1e-6), segments of S + Q clouds with their own BatchNorm statistics, chunk tails (N % 32 != 0), and the
    default form bit-identical from run to run."""
    from r3dfsseg_amd import _lib, ops, train_ops as T
    lib = _lib.load()
    rs = np.random.RandomState(7 + K)
    C = 64
    x = torch.from_numpy(rs.randn(B * N, C).astype(np.float32)).cuda()
    idx = torch.from_numpy(np.stack([[rs.permutation(N)[:K] for _ in range(N)] for _ in range(B)]).astype(np.int32)).cuda()
    conv1, conv2 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).cuda(), torch.nn.Conv2d(64, 64, 1, bias=False).cuda()
    bn1, bn2 = torch.nn.BatchNorm2d(64).cuda().train(), torch.nn.BatchNorm2d(64).cuda().train()
    with torch.no_grad():
        for bn in (bn1, bn2):
            bn.weight.copy_(torch.from_numpy(rs.uniform(-1.5, 1.5, 64).astype(np.float32)))  # max AND min winners
            bn.bias.copy_(torch.from_numpy(rs.randn(64).astype(np.float32) * 0.3))
    ec = SimpleNamespace(layer=[conv1, bn1, None, conv2, bn2])
    R = torch.from_numpy(rs.randn(B * N, 64).astype(np.float32)).cuda()
    seg = ops.SegLayout(B // (S + Q), S, Q, N)
    out = torch.empty(B * N, 64, device="cuda")
    saved = T.edgeconv_train_fwd(x, idx, ec, B, N, out, seg)
    before = lib.r3d_get_matrix_arith()
    res = {}
    try:
        for arith in (0, 1, 1):
            lib.r3d_set_matrix_arith(arith)
            dx = torch.zeros(B * N, C, device="cuda")
            grads = T.edgeconv_train_bwd(saved, R, B, N, dx)
            torch.cuda.synchronize()
            res.setdefault(arith, []).append([dx] + [g.clone() for g in grads])
    finally:
        lib.r3d_set_matrix_arith(before)
    names = ("dx", "dW1", "dg1", "db1", "dW2", "dg2", "db2")
    for a, b in zip(res[1][0], res[1][1]):
        assert torch.equal(a, b)
    worst = {n: _rel(a, b) for n, a, b in zip(names, res[1][0], res[0][0])}
    print("bf16 x 3 against fp32 core, max |diff| / max |value|:", {k: "%.1e" % v for k, v in worst.items()})
    assert max(worst.values()) < 2e-5, worst


def test_attention_dropout_mask_and_backward():
    """Dropout mask is a stateless hash: replicate it on the host, feed it to the oracle."""
    from r3dfsseg_amd import ops, _lib
    import ctypes
    B, N, p, seed = 1, 256, 0.1, 12345
    rs = np.random.RandomState(3)
    qkv = torch.from_numpy(rs.randn(B * N, 192).astype(np.float32) * 0.7).requires_grad_()
    R = torch.from_numpy(rs.randn(B * N, 64).astype(np.float32))
    row = np.arange(B * N, dtype=np.uint32)[:, None]
    key = np.arange(N, dtype=np.uint32)[None, :]
    with np.errstate(over="ignore"):
        x = row * np.uint32(0x9E3779B1) ^ key * np.uint32(0x85EBCA77) ^ np.uint32(seed)
        x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    keep = torch.from_numpy((x >= np.uint32(int(p * 4294967296.0))).astype(np.float32)) / (1 - p)
    q, k, v = qkv[:, :64], qkv[:, 64:128], qkv[:, 128:]
    attn = torch.softmax(q @ k.t(), -1) * keep
    y = attn @ v
    (y * R).sum().backward()
    lib = _lib.load()
    qg = qkv.detach().cuda()
    out = torch.empty(B * N, 64, device="cuda"); lse = torch.empty(B * N, device="cuda")
    st = ops._st()
    # effective seed = immediate + device word (the word is what a captured episode graph bumps per replay)
    sdev = torch.tensor([1000], dtype=torch.int32, device="cuda")
    aws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")  # enables the key-axis split (8 ranges here)
    _lib.check(lib.r3d_attention_fwd_train(ops._p(qg), 192, B, N, ops._p(out), 64, ops._p(lse), p, ctypes.c_uint(seed - 1000),
                                           ops._p(sdev), ops._p(aws), st))
    np.testing.assert_allclose(out.cpu().numpy(), y.detach().numpy(), atol=1e-4, rtol=1e-4)
    assert 0.08 < 1 - (keep > 0).float().mean().item() < 0.12
    dqkv = torch.empty(B * N, 192, device="cuda"); ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
    Rg = R.cuda()
    _lib.check(lib.r3d_attention_bwd(ops._p(qg), 192, B, N, ops._p(out), 64, ops._p(Rg), 64, ops._p(lse), p, ctypes.c_uint(seed),
                                     None, 1.0, ops._p(dqkv), 192, ops._p(ws), st))
    assert _rel(dqkv.cpu(), qkv.grad) < 1e-3


def _train_episode(n_way=2, k_shot=2, N=512, seed=3, noise=0.0):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=n_way, k_shot=k_shot, pc_npts=N)
    sd = S.make_state_dict(cfg, 123)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().train()
    m.att_learner.dropout.p = 0.0  # parity needs a shared mask; the mask itself is tested above
    data, _ = S.make_episode(cfg, seed, noise_ratio=noise, train=True)
    return cfg, sd, m, data


@pytest.mark.parametrize("n_way,k_shot", [(2, 2), (4, 2), (5, 2)])  # (k_shot >= 2: mpti.py:270 samples two negative shots)
def test_head_losses_and_feature_gradients(n_way, k_shot):
    """LP loss + contrastive loss and their gradients w.r.t. the features and proj, given equal features.  More than 3
    ways: the label propagation and its adjoint run as two planes of 4 label columns whose gradients add."""
    from r3dfsseg_amd import contrast, head_train
    cfg, sd, m, data = _train_episode(n_way=n_way, k_shot=k_shot)
    sx, sy, qx, qy = data[:4]
    flag = data[10]
    rs = np.random.RandomState(11)
    Sn, N = n_way * k_shot, 512
    n_q, C = qx.shape[0], n_way + 1
    sfeat = torch.from_numpy((rs.randn(Sn * N, 192) * 0.08).astype(np.float32))
    qfeat = torch.from_numpy((rs.randn(n_q * N, 192) * 0.08).astype(np.float32))
    # --- device
    sg, qg = sfeat.cuda().requires_grad_(), qfeat.cuda().requires_grad_()
    closs = contrast.per_way_contrast_loss(m, sg, sy.cuda(), flag.cuda())
    lploss = head_train.HeadLPFn.apply(sg, qg, m, sy.cuda(), qy.cuda())
    (lploss + 0.1 * closs).backward()
    assert m._head[1].stats.cpu().tolist()[0] == 1 and m._head[1].stats_bwd.cpu().tolist()[0] == 1
    # --- oracle (torch-CPU autograd through the restated head)
    so, qo = sfeat.clone().requires_grad_(), qfeat.clone().requires_grad_()
    sdr = {k: v.clone() for k, v in sd.items()}
    sdr["proj.weight"].requires_grad_(); sdr["proj.bias"].requires_grad_()
    sf4 = so.view(n_way, k_shot, N, 192).transpose(2, 3)  # (n_way, k_shot, d, N)
    c_ref = O.per_way_contrast_loss(sdr, sf4, sy, flag, 4, 0.1)
    fg_p, fg_l, _, _ = O.get_foreground_prototypes(sf4, sy, 100, C)
    bg_p, bg_l, _, _ = O.get_background_prototypes(sf4, torch.logical_not(sy), 100, C)
    protos = torch.cat((bg_p, fg_p), 0)
    Y = torch.zeros(protos.shape[0] + n_q * N, C)
    Y[:protos.shape[0]] = torch.cat((bg_l, fg_l), 0)
    node = torch.cat((protos, qo), 0)
    A = O.affinity(node, 200, 1.0)
    Z = O.label_propagate(A, Y)
    qpred = Z[protos.shape[0]:].view(-1, N, C).transpose(1, 2)
    lp_ref = torch.nn.functional.cross_entropy(qpred, qy)
    (lp_ref + 0.1 * c_ref).backward()
    assert abs(closs.item() - c_ref.item()) < 1e-4 * max(1, abs(c_ref.item())), (closs.item(), c_ref.item())
    assert abs(lploss.item() - lp_ref.item()) < 1e-4 * max(1, abs(lp_ref.item())), (lploss.item(), lp_ref.item())
    e_s, e_q = _rel(sg.grad.cpu(), so.grad), _rel(qg.grad.cpu(), qo.grad)
    e_w, e_b = _rel(m.proj.weight.grad.cpu(), sdr["proj.weight"].grad), _rel(m.proj.bias.grad.cpu(), sdr["proj.bias"].grad)
    print("head gradient errors: sfeat %.2e qfeat %.2e proj.w %.2e proj.b %.2e" % (e_s, e_q, e_w, e_b))
    assert e_s < 2e-3 and e_q < 2e-3 and e_w < 1e-3 and e_b < 1e-3


@pytest.mark.parametrize("n_way,k_shot", [(4, 2), (6, 2)])
def test_training_episode_with_more_than_three_ways(n_way, k_shot):
    """A whole training episode with 5 / 7 classes (models/mpti.py:49,58 take any n_way) through MPTI_SelfAtten.forward
    and backward: losses, the four debug metrics (mpti.py:515-568) and the feature / proj gradients against the oracle's
    head on the HIP features (201-NN lists injected, as in the full-size test)."""
    from r3dfsseg_amd import ops
    cfg, sd, m, data = _train_episode(n_way=n_way, k_shot=k_shot, N=512, seed=9, noise=0.2)
    ep = [t.cuda() for t in data]
    sx, sy, qx, qy, gsy, gqy, flag = data[0], data[1], data[2], data[3], data[6], data[7], data[10]
    N, Sn = 512, n_way * k_shot
    m._trace = {}
    out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
            lp_iters=m.lp_max_iter)
    (out[1] + 0.1 * out[2]).backward()
    assert m.lp_converged(backward=True)
    assert out[0].shape == (qx.shape[0], n_way + 1, N)
    tr = m._trace
    sfeat, qfeat = tr["sfeat"], tr["qfeat"]
    so = sfeat.detach().cpu().reshape(Sn, N, -1).transpose(1, 2).contiguous().requires_grad_()
    qo = qfeat.detach().cpu().reshape(qx.shape[0], N, -1).transpose(1, 2).contiguous().requires_grad_()
    sdh = {k_: v.clone() for k_, v in sd.items()}
    sdh["proj.weight"].requires_grad_(); sdh["proj.bias"].requires_grad_()
    hb = m._head[1]
    n_proto, n = int(hb.desc[ops.HD_N_PROTO].item()), int(hb.desc[ops.HD_N_NODES].item())
    nbr_hip = tr["nbr"].reshape(hb.n_cap, hb.kp1)[:n].cpu().to(torch.int64)
    assert torch.equal(nbr_hip, O.knn_l2(hb.nodes[:n].cpu(), hb.kp1))
    ref, aux = O.mpti_head(sdh, cfg, so, qo, sx, sy, qy, gt_support_y=gsy, gt_query_y=gqy, train=True, support_flag=flag,
                           nbr_override=nbr_hip, return_aux=True)
    assert aux["n_proto"] == n_proto
    (ref[1] + 0.1 * ref[2]).backward()
    assert abs(out[1].item() - ref[1].item()) <= 1e-4 * max(1.0, abs(ref[1].item())), (out[1].item(), ref[1].item())
    assert abs(out[2].item() - ref[2].item()) <= 1e-4 * max(1.0, abs(ref[2].item())), (out[2].item(), ref[2].item())
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), ref[0].detach().numpy(), atol=1e-4)
    e_s = _rel(sfeat.grad.cpu(), so.grad.transpose(1, 2).reshape(Sn * N, -1))
    e_q = _rel(qfeat.grad.cpu(), qo.grad.transpose(1, 2).reshape(qx.shape[0] * N, -1))
    e_w = _rel(m.proj.weight.grad.cpu(), sdh["proj.weight"].grad)
    assert max(e_s, e_q, e_w) <= 2e-3, (e_s, e_q, e_w)
    for i, (a, b) in enumerate(zip(out[3:], ref[3:])):
        assert abs(float(a) - float(b)) <= 2e-3, (i, float(a), float(b))


def test_learner_train_step_runs_and_descends():
    """MPTILearner_V3.train (mpti_learner.py:50-78): 8-tuple, finite, loss decreases over a few steps."""
    from r3dfsseg_amd.mpti_learner import MPTILearner_V3
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512, pretrain_checkpoint_path="synthetic", model_checkpoint_path=None,
                     lr=1e-3, step_size=5000, gamma=0.5)
    L = MPTILearner_V3(SimpleNamespace(**cfg), mode="train")
    data, _ = S.make_episode(cfg, 5, noise_ratio=0.5, train=True)
    data = [t.cuda() for t in data]
    losses = []
    for it in range(6):
        out = L.train(data, None)
        assert len(out) == 8
        loss, lp, con, acc = out[0], out[1], out[2], out[3]
        assert torch.isfinite(loss) and torch.isfinite(lp) and torch.isfinite(con) and 0.0 <= acc <= 1.0
        losses.append(loss.item())
    print("train losses:", losses)
    assert losses[-1] < losses[0]
    assert int(L.model.encoder.conv.layer[1].num_batches_tracked.item()) == 12  # two getFeatures calls per step


@pytest.mark.parametrize("N", [512, 40])
def test_one_launch_sequence_equals_two_getfeatures_calls(N):
    """mpti.py:434,436 call getFeatures twice; the training path runs both calls through ONE launch sequence with the
    BatchNorm statistics of the two calls kept apart (segments).  Against two separate calls (support clouds alone, query
    clouds alone): features of every BatchNorm-only path and the running statistics bit for bit (a segment's reductions
    do not depend on what it is batched with; N = 40: segments that do not end on the GEMM tiles), losses to rounding,
    gradients equal up to LeakyReLU kinks."""
    from r3dfsseg_amd import contrast, head_train, train_ops as T
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=N)
    data, _ = S.make_episode(cfg, seed=11, noise_ratio=0.5, train=True)
    ep = [t.cuda() for t in data]
    Sn = cfg["n_way"] * cfg["k_shot"]
    res = {}
    for joint in (False, True):
        m = MPTI_SelfAtten(SimpleNamespace(**cfg))
        m.load_state_dict(S.make_state_dict(cfg, 123))
        m.cuda().train()
        m.att_learner.dropout.p = 0.0
        m._trace = {}
        if joint:
            out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
                    lp_iters=m.lp_max_iter)
            lp, cl = out[1], out[2]
            sfeat, qfeat = m._trace["sfeat"], m._trace["qfeat"]
        else:
            m._lp_force = True
            sfeat = T.get_features_train(m, ep[0].reshape(Sn, cfg["pc_in_dim"], N), seed=2)
            qfeat = T.get_features_train(m, ep[2], seed=3)
            cl = contrast.per_way_contrast_loss(m, sfeat, ep[1], ep[10])
            lp = head_train.HeadLPFn.apply(sfeat, qfeat, m, ep[1], ep[3])
        (lp + 0.1 * cl).backward()
        assert m.lp_converged(backward=True)
        res[joint] = dict(lp=lp.item(), cl=cl.item(), sfeat=sfeat.detach().clone(), qfeat=qfeat.detach().clone(),
                          grads={n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                          stats={n: b.clone() for n, b in m.named_buffers()})
    a, b = res[False], res[True]
    for n_ in a["stats"]:
        assert torch.equal(a["stats"][n_], b["stats"][n_]), n_
    for f in ("sfeat", "qfeat"):  # level-1 and BaseLearner columns see BatchNorm only; the attention columns depend on the
        # key-axis split, which follows the number of clouds in the launch
        assert torch.equal(a[f][:, :64], b[f][:, :64]) and torch.equal(a[f][:, 128:], b[f][:, 128:]), f
        assert (a[f][:, 64:128] - b[f][:, 64:128]).abs().max().item() < 1e-5
    assert abs(a["lp"] - b["lp"]) <= 1e-5 * max(1.0, abs(a["lp"])) and abs(a["cl"] - b["cl"]) <= 1e-5 * max(1.0, abs(a["cl"]))
    for n_ in a["grads"]:
        assert _rel(b["grads"][n_], a["grads"][n_]) <= 2e-3, (n_, _rel(b["grads"][n_], a["grads"][n_]))


@pytest.mark.gpu
@pytest.mark.parametrize("C", [64, 128, 256, 512])
@pytest.mark.parametrize("E,S,Q,N", [(3, 3, 1, 100), (2, 2, 1, 96)])  # (N = 100: segments that end inside a workgroup's rows)
def test_vectorised_batchnorm_passes_give_the_scalar_kernels_bits(C, E, S, Q, N):
    """r3d_bn_bwd_apply_v4 / r3d_colpartial_v4 (16 bytes per lane, constants formed once per workgroup) against the
    4-bytes-per-lane kernels they replace -- selected by giving the same matrices rows that are not 16-byte aligned
    (leading dimension C + 1): every sum and every element bit for bit."""
    from r3dfsseg_amd import ops, train_ops as T
    seg = ops.SegLayout(E, S, Q, N)
    M = seg.M
    g = torch.Generator().manual_seed(C + N)
    dev = torch.device("cuda")
    Z = torch.randn(M, C, generator=g).to(dev)
    DY = torch.randn(M, C, generator=g).to(dev)
    bn = T.BNVec(seg.n_seg, C, dev)
    bn.t.copy_(torch.randn(seg.n_seg, 4, C, generator=g).to(dev))
    bn.t[:, 3].abs_().add_(0.5)

    def unaligned(t):
        p = torch.empty(M, C + 1, device=dev)
        p[:, :C] = t
        return p[:, :C]
    for act in (0, 2):
        s_fast = T.colstats(Z, C, seg, mode=1, DY=DY, bn=bn, act=act)
        s_slow = T.colstats(unaligned(Z), C, seg, mode=1, DY=unaligned(DY), bn=bn, act=act)
        assert torch.equal(s_fast.view(torch.int32), s_slow.view(torch.int32)), "backward column sums differ (C %d)" % C
        f_fast, f_slow = T.colstats(Z, C, seg, mode=0), T.colstats(unaligned(Z), C, seg, mode=0)
        assert torch.equal(f_fast.view(torch.int32), f_slow.view(torch.int32)), "forward column sums differ (C %d)" % C
        d_fast = T.bn_bwd_apply(Z, DY, bn, act, s_fast, seg.counts(), seg)
        d_slow = T.bn_bwd_apply(unaligned(Z), unaligned(DY), bn, act, s_fast, seg.counts(), seg, out=unaligned(torch.zeros(M, C, device=dev)))
        assert torch.equal(d_fast.view(torch.int32), d_slow.contiguous().view(torch.int32)), "dz differs (C %d, act %d)" % (C, act)
