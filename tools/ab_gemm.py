"""Point-wise GEMM and weight-gradient GEMM: fp32 MFMA against bf16 x 3 -- accuracy against float64 and time per call."""
import sys

import torch

sys.path.insert(0, ".")
from r3dfsseg_amd import _lib, ops, train_ops as T
from r3dfsseg_amd.ops import _p, _st

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
torch.manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("point-wise GEMM  Out = act(scale X W^T + shift), M = %d" % M)
for K, Co in ((64, 128), (192, 512), (512, 256), (256, 128), (128, 64), (256, 192), (128, 64), (512, 192), (256, 512), (64, 64)):
    X = torch.randn(M, K, device="cuda")
    W = torch.randn(Co, K, device="cuda") / K ** 0.5
    sc, sh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda")
    want = torch.relu((X.double() @ W.double().t()) * sc.double() + sh.double())
    res = []
    for mode in (0, 1):
        _lib.check(lib.r3d_set_matrix_arith(mode))
        out = ops.pointwise_conv(X, W, sc, sh, ops.ACT_RELU)
        err = ((out.double() - want).abs().max() / want.abs().max()).item()
        t = timed(lambda: ops.pointwise_conv(X, W, sc, sh, ops.ACT_RELU))
        # training form: raw z and its column sums
        z = torch.empty(M, Co, device="cuda")
        sums = torch.empty(2 * Co, device="cuda")
        ws = torch.empty(lib.r3d_pointwise_conv_stats_ws_words(M, Co), device="cuda")
        _lib.check(lib.r3d_pointwise_conv_stats(_p(X), K, _p(W), M, K, Co, _p(z), Co, _p(sums), _p(ws), _st()))
        zz = X.double() @ W.double().t()
        serr = max(((sums[:Co].double() - zz.sum(0)).abs().max() / zz.sum(0).abs().max()).item(),
                   ((sums[Co:].double() - (zz * zz).sum(0)).abs().max() / (zz * zz).sum(0).abs().max()).item())
        res.append((err, serr, t))
    _lib.check(lib.r3d_set_matrix_arith(1))
    gf = 2.0 * M * K * Co / 1e9
    print("  K %3d Co %3d  fp32: %6.1f us (%5.1f TF/s) err %.1e stats %.1e | bf16x3: %6.1f us (%5.1f TF/s) err %.1e stats %.1e" % (
        K, Co, res[0][2], gf / res[0][2] * 1e-3 * 1e3, res[0][0], res[0][1], res[1][2], gf / res[1][2] * 1e-3 * 1e3, res[1][0], res[1][1]))

print("weight-gradient GEMM  A^T B")
for Ca, Cb in ((128, 64), (512, 192), (256, 512), (128, 256), (64, 128), (192, 256), (64, 64), (128, 9)):
    A = torch.randn(M, Ca, device="cuda")
    Bm = torch.randn(M, Cb, device="cuda")
    want = A.double().t() @ Bm.double()
    res = []
    for mode in (0, 1):
        _lib.check(lib.r3d_set_matrix_arith(mode))
        out = T.gemm_tn(A, Bm)
        err = ((out.double() - want).abs().max() / want.abs().max()).item()
        res.append((err, timed(lambda: T.gemm_tn(A, Bm))))
    _lib.check(lib.r3d_set_matrix_arith(1))
    gf = 2.0 * M * Ca * Cb / 1e9
    print("  Ca %3d Cb %3d  fp32: %6.1f us (%5.1f TF/s) err %.1e | bf16x3: %6.1f us (%5.1f TF/s) err %.1e" % (
        Ca, Cb, res[0][1], gf / res[0][1], res[0][0], res[1][1], gf / res[1][1], res[1][0]))
