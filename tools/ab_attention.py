"""Attention forward / backward: fp32 MFMA against bf16 x 3 -- accuracy against a float64 torch reference and
time per call (HIP events, 20 calls)."""
import sys

import torch

sys.path.insert(0, ".")
from r3dfsseg_amd import _lib, ops
from r3dfsseg_amd.ops import _p, _st

lib = _lib.load()
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 2048)
torch.manual_seed(0)
qkv = torch.randn(B * N, 192, device="cuda")
qkv[:, :64] *= 0.5
q, k, v = (qkv[:, 64 * i:64 * (i + 1)].double().view(B, N, 64) for i in range(3))
P = torch.softmax(q @ k.transpose(1, 2), -1)
want = (P @ v).reshape(B * N, 64)
dO = torch.randn(B * N, 64, device="cuda")
dOd = dO.double().view(B, N, 64)
dV = P.transpose(1, 2) @ dOd
dP = dOd @ v.transpose(1, 2)
dS = P * (dP - (dP * P).sum(-1, keepdim=True))
dQ, dK = dS @ k, dS.transpose(1, 2) @ q
want_d = torch.cat((dQ, dK, dV), -1).reshape(B * N, 192)
ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")


def run(mode):
    _lib.check(lib.r3d_set_matrix_arith(mode))
    out = torch.empty(B * N, 64, device="cuda")
    lse = torch.empty(B * N, device="cuda")
    dqkv = torch.empty(B * N, 192, device="cuda")

    def fwd():
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.0, 0, None, _p(ws), _st()))

    def bwd():
        _lib.check(lib.r3d_attention_bwd(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.0, 0, None, 1.0, _p(dqkv),
                                         192, _p(ws), _st()))

    res = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    err = ((out.double() - want).abs().max() / want.abs().max()).item()
    errd = [((dqkv[:, 64 * i:64 * (i + 1)].double() - want_d[:, 64 * i:64 * (i + 1)]).abs().max()
             / want_d[:, 64 * i:64 * (i + 1)].abs().max()).item() for i in range(3)]
    return res, err, errd, out.clone(), dqkv.clone()


r0 = run(0)
r1 = run(1)
for name, r in (("fp32 MFMA", r0), ("bf16 x 3", r1)):
    print("%-10s fwd %.1f us  bwd %.1f us   max error / max |.| against float64: out %.2e  dq %.2e dk %.2e dv %.2e" % (
        name, r[0]["fwd"], r[0]["bwd"], r[1], *r[2]))
print("bf16 x 3 against fp32 MFMA: out %.2e  dqkv %.2e" % ((r0[3] - r1[3]).abs().max().item(), (r0[4] - r1[4]).abs().max().item()))
