"""Error of the point-wise GEMM and the weight-gradient GEMM against float64 in each arithmetic: fp32 matrix core, bf16 x 3.
(The run recorded in profiles/r03_experiments.md also had a form with the two middle x low cross terms, eight products:
no more accurate -- the fp32 accumulation, not the dropped 2^-24 terms, sets the error -- and removed.)  Reported relative to sum |x||w| (the scale rounding errors follow): max, rms and
the MEAN signed error (a bias shows truncating accumulation)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import _lib, ops, train_ops as T  # noqa: E402

lib = _lib.load()
torch.manual_seed(0)
for (M, K, Co) in [(8192, 192, 512), (8192, 512, 256), (8192, 64, 128)]:
    x = torch.randn(M, K); W = torch.randn(Co, K) / K ** 0.5
    x = torch.relu(x) + 0.1 * torch.randn(M, K)  # activations are mostly positive: sums without cancellation
    ref = x.double() @ W.double().t(); scale = x.double().abs() @ W.double().abs().t()
    xd, Wd = x.cuda(), W.cuda()
    for name, arith, mask in (("fp32", 0, 3), ("bx3", 1, 3)):
        lib.r3d_set_matrix_arith(arith); lib.r3d_debug_set_gemm_bx3(mask)
        got = ops.pointwise_conv(xd, Wd, None, None, 0).cpu().double()
        e = (got - ref) / scale
        er = (got - ref) / ref.abs().clamp_min(1e-30)
        print("pointwise %5d x %3d -> %3d  %-10s  max %.2e  rms %.2e  mean %+.2e   | rel to |ref|: rms %.2e mean %+.2e" % (
            M, K, Co, name, e.abs().max(), e.pow(2).mean().sqrt(), e.mean(), er.pow(2).mean().sqrt(), er.mean()))
for (M, Ca, Cb) in [(262144, 128, 64), (262144, 256, 128)]:
    A = torch.randn(M, Ca); B = torch.relu(torch.randn(M, Cb))
    ref = A.double().t() @ B.double(); scale = A.double().abs().t() @ B.double().abs()
    Ad, Bd = A.cuda(), B.cuda()
    for name, arith, mask in (("fp32", 0, 3), ("bx3", 1, 3)):
        lib.r3d_set_matrix_arith(arith); lib.r3d_debug_set_gemm_bx3(mask)
        got = T.gemm_tn(Ad, Bd).cpu().double()
        e = (got - ref) / scale
        print("gemm_tn %6d x %3d x %3d  %-10s  max %.2e  rms %.2e  mean %+.2e   | rel to max|ref| %.2e" % (
            M, Ca, Cb, name, e.abs().max(), e.pow(2).mean().sqrt(), e.mean(), (got - ref).abs().max() / ref.abs().max()))
