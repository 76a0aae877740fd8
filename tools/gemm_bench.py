"""Per-shape timing of the index-free GEMMs in both matrix arithmetics (fp32 core: gemm.hip / train_ops.hip; bf16 core in
three pieces: gemm_bx3.hip) at the row count of a 32-episode step of workload S.
    python tools/gemm_bench.py [--rows 262144]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import _lib, ops, train_ops as T  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    args = ap.parse_args()
    M = args.rows
    lib = _lib.load()
    shapes = [(64, 128), (192, 512), (512, 256), (256, 128), (128, 64), (256, 192), (512, 192), (256, 512), (128, 256), (64, 128),
              (128, 64), (64, 64)]
    print("pointwise conv  (M = %d)     fp32 us   bx3 us   speedup   bx3 TFLOP/s (fp32-equivalent)   bx3 GB/s" % M)
    for K, Co in shapes:
        x = torch.randn(M, K, device="cuda")
        W = torch.randn(Co, K, device="cuda") / K ** 0.5
        out = torch.empty(M, Co, device="cuda")
        t = []
        for arith in (0, 1):
            _lib.check(lib.r3d_set_matrix_arith(arith))
            t.append(timeit(lambda: ops.pointwise_conv(x, W, None, None, 0, out=out)))
        fl = 2.0 * M * K * Co
        by = 4.0 * M * (K + Co)
        print("  %4d -> %4d              %8.1f %8.1f   %5.2f      %6.1f                        %6.0f" %
              (K, Co, t[0], t[1], t[0] / t[1], fl / t[1] * 1e-6, by / t[1] * 1e-3))
    print("gemm_tn A^T B (M = %d)" % M)
    for Ca, Cb in [(512, 192), (256, 512), (128, 256), (64, 128), (192, 256), (128, 64), (128, 9)]:
        A = torch.randn(M, Ca, device="cuda")
        B = torch.randn(M, Cb, device="cuda")
        t = []
        for arith in (0, 1):
            _lib.check(lib.r3d_set_matrix_arith(arith))
            t.append(timeit(lambda: T.gemm_tn(A, B)))
        fl = 2.0 * M * Ca * Cb
        by = 4.0 * M * (Ca + Cb)
        print("  %4d x %4d              %8.1f %8.1f   %5.2f      %6.1f                        %6.0f" %
              (Ca, Cb, t[0], t[1], t[0] / t[1], fl / t[1] * 1e-6, by / t[1] * 1e-3))


if __name__ == "__main__":
    main()
