"""Do PyTorch's own elementwise kernels (gradient accumulation, Adam, sums) compute the same values while the bf16 x 3
attention kernels run on another stream?  (They are compiled by somebody else and may contain packed fp32 arithmetic.)"""
import sys
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p
lib = _lib.load()
B, N = 12, 2048
torch.manual_seed(0)
qkv = torch.randn(B * N, 192, device="cuda"); dO = torch.randn(B * N, 64, device="cuda")
aws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
out = torch.empty(B * N, 64, device="cuda"); lse = torch.empty(B * N, device="cuda"); dqkv = torch.empty(B * N, 192, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
n = 4 * 1024 * 1024
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.rand(n, device="cuda") + 0.5
params = [torch.randn(s, device="cuda") for s in (64 * 64, 128 * 64, 512 * 192, 256 * 512, 64, 128, 192 * 256)]
grads = [torch.randn_like(p) for p in params]


def work():
    res = []
    with torch.cuda.stream(sa):
        res.append(a + b)
        res.append(a * b + c)
        res.append(torch.addcmul(a, b, c, value=0.37))
        res.append((a * a).sum(dtype=torch.float32).reshape(1))
        res.append(torch.sqrt(c) / (b.abs() + 1e-3))
        acc = [p.clone() for p in params]
        torch._foreach_add_(acc, grads)
        torch._foreach_mul_(acc, 0.9)
        torch._foreach_addcdiv_(acc, grads, [g.abs() + 1.0 for g in grads], value=-0.01)
        res += acc
        m = a.view(2048, 2048)
        res.append(m.sum(0))
        res.append(torch.cat((a[:1000], b[:1000])) * 2.0)
    return res


def load(k):
    with torch.cuda.stream(sb):
        for _ in range(k):
            _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.1, 7, None, _p(aws), sb.cuda_stream))
            _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.1, 7, None, 0.125, _p(dqkv), 192,
                                                _p(aws), 1, sb.cuda_stream))


ref = work()
torch.cuda.synchronize()
again = work()
torch.cuda.synchronize()
print("alone twice identical:", all(torch.equal(x, y) for x, y in zip(ref, again)))
for mode, name in ((1, "bf16 x 3 attention"), (0, "fp32 attention")):
    _lib.check(lib.r3d_set_matrix_arith(mode))
    bad = 0
    tot = 0
    for rep in range(30):
        load(30)
        got = work()
        torch.cuda.synchronize()
        for i, (x, y) in enumerate(zip(ref, got)):
            tot += 1
            if not torch.equal(x, y):
                bad += 1
                if bad <= 5:
                    d = (x - y).abs()
                    print("  rep %d result %d differs: %d entries, max %.3e" % (rep, i, int((d > 0).sum()), d.max().item()))
    print("beside %s: %d of %d torch results differ" % (name, bad, tot))
_lib.check(lib.r3d_set_matrix_arith(1))
