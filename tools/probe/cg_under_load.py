"""Label-propagation solves on one stream while another stream keeps the memory system busy (bf16 x 3 attention, fp32
attention, or a copy loop): the iteration count and the result of every solve must not depend on what runs beside it."""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p
lib = _lib.load()
n, kp1, D = 4396, 201, 192
torch.manual_seed(0)
cent = torch.randn(3, D, device="cuda") * 0.5
x = (cent[torch.randint(0, 3, (n,), device="cuda")] + torch.randn(n, D, device="cuda") * 0.12).contiguous()
Y = torch.zeros(n, 4, device="cuda"); Y[torch.arange(300), torch.randint(0, 3, (300,))] = 1
nd = torch.tensor([n], device="cuda", dtype=torch.int32); npd = torch.tensor([300], device="cuda", dtype=torch.int32)
norm = torch.empty(lib.r3d_knn_norm_ws_words(1, n), device="cuda"); cm = torch.empty(D * lib.r3d_cm_pitch(n), device="cuda")
nbr = torch.empty(n, kp1, device="cuda", dtype=torch.int32); st = torch.zeros(1, device="cuda", dtype=torch.int32)
_lib.check(lib.r3d_knn_topk(_p(x), D, None, 1, n, D, kp1, 1, _p(nd), _p(norm), _p(cm), _p(nbr), None, _p(st), None))
ws = torch.empty(lib.r3d_lp_ws_words(n, kp1), device="cuda", dtype=torch.int32)
B, N = 12, 2048
qkv = torch.randn(B * N, 192, device="cuda"); dO = torch.randn(B * N, 64, device="cuda")
aws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
out = torch.empty(B * N, 64, device="cuda"); lse = torch.empty(B * N, device="cuda"); dqkv = torch.empty(B * N, 192, device="cuda")
big = torch.empty(64 * 1024 * 1024, device="cuda"); big2 = torch.empty_like(big)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def solves(k):
    res = []
    with torch.cuda.stream(sa):
        for _ in range(k):
            Z = torch.empty(n, 4, device="cuda"); stats = torch.zeros(2, device="cuda", dtype=torch.int32)
            _lib.check(lib.r3d_label_propagate(_p(x), D, D, _p(nbr), kp1, _p(Y), _p(nd), _p(npd), n, 1.0, 0.99, 200, 1e-6, _p(Z),
                                               _p(ws), ws.numel(), _p(stats), sa.cuda_stream))
            res.append((Z, stats))
    return res


def load(kind, k):
    with torch.cuda.stream(sb):
        for _ in range(k):
            if kind == "copy":
                big2.copy_(big)
            else:
                _lib.check(lib.r3d_set_matrix_arith(1 if kind == "bx3" else 0))
                _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.1, 7, None, _p(aws), sb.cuda_stream))
                _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.1, 7, None, 0.125, _p(dqkv),
                                                    192, _p(aws), 1, sb.cuda_stream))


sink = torch.zeros(1, device="cuda", dtype=torch.int32)
ref = solves(3)
torch.cuda.synchronize()
Zref, sref = ref[0][0].clone(), ref[0][1].tolist()
print("alone: (converged, iterations) %s, repeatable %s" % (sref, all(torch.equal(r[0], Zref) for r in ref)))
for kind in ("copy", "fp32", "bx3"):
    bad = 0
    its = []
    for rep in range(6):
        load(kind, 40)
        r = solves(60)
        torch.cuda.synchronize()
        for Z, s in r:
            its.append(s.tolist()[1])
            if s.tolist() != sref or not torch.equal(Z, Zref):
                bad += 1
    print("beside %-5s: %d of %d solves differ from the solve alone; iterations min %d max %d" % (kind, bad, len(its), min(its), max(its)))
_lib.check(lib.r3d_set_matrix_arith(1))

# stale LDS: poison the chip's LDS, then solve alone
for pat in (0xffffffff, 0x7f800000, 0x3f803f80, 0x00000000):
    bad = 0
    for rep in range(20):
        with torch.cuda.stream(sa):
            _lib.check(lib.r3d_debug_poison_lds(pat, _p(sink), sa.cuda_stream))
        r = solves(1)
        torch.cuda.synchronize()
        Z, s_ = r[0]
        if s_.tolist() != sref or not torch.equal(Z, Zref):
            bad += 1
            last = (s_.tolist(), bool(torch.isfinite(Z).all()), (Z - Zref).abs().max().item())
    print("LDS poisoned with %08x: %d of 20 solves differ%s" % (pat, bad, "  e.g. %s" % (last,) if bad else ""))
