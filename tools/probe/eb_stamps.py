"""Phase clocks of two waves of r3d_edgeconv_bwd1_bx3_kernel (library built with -DEB_STAMPS by this script) at the headline
size: 384 clouds of 2048 points, K = 20.  usage (GPU box): python tools/probe/eb_stamps.py"""
import ctypes
import os
import subprocess
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "r3dfsseg_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "r3dfsseg_amd"))
import build as B_  # noqa: E402  (the library's own flags)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B_.FLAGS + ["-DEB_STAMPS", "-c", os.path.join(src, "edgeconv_train.hip"), "-o", "/tmp/ect_st.o"])
objs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith(".o") and f != "edgeconv_train.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libeb_st.so", "/tmp/ect_st.o"] + objs)
os.environ["R3D_LIB"] = "/tmp/libeb_st.so"
import numpy as np  # noqa: E402
import torch  # noqa: E402
from r3dfsseg_amd import ops, train_ops as T  # noqa: E402

L = ctypes.CDLL("/tmp/libeb_st.so")
Bc, N, C, K = int(os.environ.get("EB_CLOUDS", 384)), 2048, 64, 20
rs = np.random.RandomState(1)
x = torch.from_numpy(rs.randn(Bc * N, C).astype(np.float32)).cuda()
idx = torch.from_numpy(rs.randint(0, N, (Bc, N, K)).astype(np.int32)).cuda()
conv1, conv2 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).cuda(), torch.nn.Conv2d(64, 64, 1, bias=False).cuda()
bn1, bn2 = torch.nn.BatchNorm2d(64).cuda().train(), torch.nn.BatchNorm2d(64).cuda().train()
ec = SimpleNamespace(layer=[conv1, bn1, None, conv2, bn2])
R = torch.from_numpy(rs.randn(Bc * N, 64).astype(np.float32)).cuda()
seg = ops.SegLayout(Bc // 12, 10, 2, N)
out = torch.empty(Bc * N, 64, device="cuda")
saved = T.edgeconv_train_fwd(x, idx, ec, Bc, N, out, seg)
names = ["h1 + cut + requests", "wait B1", "GEMM1", "dz2 + cut", "wait B2", "GEMM2 + GEMM3", "wait B3", "epilogue + scan"]
for rep in range(3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        dx = torch.zeros(Bc * N, C, device="cuda")
        T.edgeconv_train_bwd(saved, R, Bc, N, dx)
    ev1.record()
    torch.cuda.synchronize()
    o = (ctypes.c_ulonglong * 32)()
    assert L.r3d_edgeconv_bwd_debug_read(o) == 0
    for wv in range(2):
        t = list(o)[16 * wv:16 * wv + 16]
        tiles = max(1, t[10])
        mhz = t[8] / max(1, t[9]) * 100.0
        print("wave %d: grid %d  tiles %d  clock %.0f MHz  kernel %d cycles = %.0f per tile  (3 backward calls: %.2f ms)" % (
            3 * wv, t[11], tiles, mhz, t[8], t[8] / tiles, ev0.elapsed_time(ev1)))
        print("   " + "  |  ".join("%s %.0f" % (n, t[i] / tiles) for i, n in enumerate(names)))
