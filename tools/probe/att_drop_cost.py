"""What the dropout decision costs the three attention kernels at the batched step's shape (384 clouds x 2048 points):
device time of forward / backward with p = 0.1 against p = 0 (the DROP = false instantiations: same kernels without the
hash).  python tools/probe/att_drop_cost.py [B N reps]"""
import sys
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p, _st
lib = _lib.load()
B, N, reps = (int(a) for a in (sys.argv[1:4] + ["384", "2048", "5"][len(sys.argv) - 1:]))
torch.manual_seed(1)
qkv = torch.randn(B * N, 192, device="cuda")
dO = torch.randn(B * N, 64, device="cuda")
seed_dev = torch.tensor([12345], device="cuda", dtype=torch.int32)
ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
out = torch.empty(B * N, 64, device="cuda"); lse = torch.empty(B * N, device="cuda")
dqkv = torch.empty(B * N, 192, device="cuda")
_lib.check(lib.r3d_set_matrix_arith(1))
for p in (0.1, 0.0, 0.1, 0.0):
    tf = tb = 0.0
    for rep in range(reps + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), p, 7, _p(seed_dev), _p(ws), _st()))
        e[1].record()
        _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), p, 7, _p(seed_dev), 0.125,
                                            _p(dqkv), 192, _p(ws), 1, _st()))
        e[2].record()
        torch.cuda.synchronize()
        if rep:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print("p = %.1f: forward %.3f ms  backward %.3f ms   (B %d N %d, mean of %d)" % (p, tf / reps, tb / reps, B, N, reps))
