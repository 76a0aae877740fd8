"""Phase clocks of one wave of r3d_attention_fwd_bx3_kernel (library built with -DATT_STAMPS by this script).
usage (GPU box): python tools/probe/att_stamps.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "r3dfsseg_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                       "-DATT_STAMPS", "-c", os.path.join(src, "attention.hip"), "-o", "/tmp/att_st.o"])
objs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith(".o") and f != "attention.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libatt_st.so",
                       "/tmp/att_st.o"] + objs)
os.environ["R3D_LIB"] = "/tmp/libatt_st.so"
import torch
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p, _st

lib = _lib.load()
L = ctypes.CDLL("/tmp/libatt_st.so")
B, N = 12, 2048
qkv = torch.randn(B * N, 192, device="cuda")
out = torch.empty(B * N, 64, device="cuda")
lse = torch.empty(B * N, device="cuda")
ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
_lib.check(lib.r3d_set_matrix_arith(1))
for rep in range(4):
    for _ in range(10):
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.0, 0, None, _p(ws), _st()))
    torch.cuda.synchronize()
    o = (ctypes.c_ulonglong * 16)()
    assert L.r3d_attention_debug_read(o) == 0
    t = list(o)
    tiles = max(1, t[8])
    mhz = t[6] / max(1, t[7]) * 100.0
    names = ["issue loads", "block A (S next + softmax + cut 0)", "rescale + block B (PV + cut 1)", "LDS stores", "barrier", "loop edge"]
    print("tiles %d  wave clock %.0f MHz  total %d cycles = %.0f per tile" % (tiles, mhz, t[6], t[6] / tiles))
    print("   " + "  |  ".join("%s %.0f" % (n, t[i] / tiles) for i, n in enumerate(names)))
