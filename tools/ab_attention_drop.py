"""bf16 x 3 attention against the fp32 kernels with dropout on, at the training shapes, seed in device memory; also the
workspace-reuse form of the backward, and repeated calls (a race would show as run-to-run differences)."""
import sys
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p, _st
lib = _lib.load()
torch.manual_seed(1)
for B, N in ((12, 2048), (10, 2048), (2, 2048)):
    qkv = torch.randn(B * N, 192, device="cuda") * 2.0
    dO = torch.randn(B * N, 64, device="cuda")
    seed_dev = torch.tensor([12345], device="cuda", dtype=torch.int32)
    res = {}
    for mode in (0, 1):
        _lib.check(lib.r3d_set_matrix_arith(mode))
        outs = []
        for rep in range(4):
            ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
            out = torch.empty(B * N, 64, device="cuda"); lse = torch.empty(B * N, device="cuda")
            dqkv = torch.empty(B * N, 192, device="cuda")
            _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.1, 7, _p(seed_dev), _p(ws), _st()))
            _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.1, 7, _p(seed_dev), 0.125,
                                                _p(dqkv), 192, _p(ws), 1 if rep % 2 else 0, _st()))
            torch.cuda.synchronize()
            outs.append((out.clone(), lse.clone(), dqkv.clone()))
        same = all(torch.equal(outs[0][i], o[i]) for o in outs[1:] for i in range(3))
        res[mode] = outs[0]
        print("B %d mode %d: 4 runs bit-identical: %s  finite: %s" % (B, mode, same, all(torch.isfinite(t).all().item() for t in outs[0])))
    for i, name in enumerate(("out", "lse", "dqkv")):
        a, b = res[0][i], res[1][i]
        print("   %s: max |fp32 - bx3| / max|fp32| = %.2e" % (name, ((a - b).abs().max() / a.abs().max()).item()))
_lib.check(lib.r3d_set_matrix_arith(1))
