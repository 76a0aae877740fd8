"""Steady-state time of the point-wise GEMM and the weight-gradient GEMM at the shapes of workload S (training)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops, train_ops as T
dev = "cuda"
def timeit(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("%-6s %-5s %-5s %9s %8s   %9s %8s" % ("M", "K", "Co", "fwd us", "TF/s", "tn us", "TF/s"))
for M in (20480, 4096):
    for K, Co in ((9, 128), (64, 128), (192, 512), (512, 256), (256, 128), (128, 64), (256, 192), (128, 64), (64, 128),
                  (128, 9), (128, 256), (256, 512), (512, 192), (192, 256)):
        X = torch.randn(M, K, device=dev); W = torch.randn(Co, K, device=dev) * 0.1
        out = torch.empty(M, Co, device=dev)
        t = timeit(lambda: ops.pointwise_conv(X, W, out=out) if "out" in ops.pointwise_conv.__code__.co_varnames else ops.pointwise_conv(X, W))
        dz = torch.randn(M, Co, device=dev)
        t2 = timeit(lambda: T.gemm_tn(dz, X))
        fl = 2.0 * M * K * Co
        print("%-6d %-5d %-5d %9.1f %8.1f   %9.1f %8.1f" % (M, K, Co, t, fl / t / 1e6, t2, fl / t2 / 1e6))
