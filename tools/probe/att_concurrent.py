"""bf16 x 3 attention forward + backward on 6 HIP streams at once (own buffers each, same inputs), repeated: every
result must equal the single-stream result bit for bit."""
import sys
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib
from r3dfsseg_amd.ops import _p
lib = _lib.load()
B, N, S = 12, 2048, 6
torch.manual_seed(3)
qkv = torch.randn(B * N, 192, device="cuda")
dO = torch.randn(B * N, 64, device="cuda")
seed_dev = torch.tensor([5], device="cuda", dtype=torch.int32)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.check(lib.r3d_set_matrix_arith(mode))


def bufs():
    return dict(ws=torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda"), out=torch.empty(B * N, 64, device="cuda"),
                lse=torch.empty(B * N, device="cuda"), dqkv=torch.empty(B * N, 192, device="cuda"))


def run(b, stream):
    st = torch.cuda.current_stream().cuda_stream if stream is None else stream.cuda_stream
    _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(b["out"]), 64, _p(b["lse"]), 0.1, 7, _p(seed_dev), _p(b["ws"]), st))
    _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(b["out"]), 64, _p(dO), 64, _p(b["lse"]), 0.1, 7, _p(seed_dev), 0.125,
                                        _p(b["dqkv"]), 192, _p(b["ws"]), 1, st))


ref = bufs()
run(ref, None)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(S)]
bs = [bufs() for _ in range(S)]
noise = torch.randn(4096, 4096, device="cuda")
bad = 0
for rep in range(30):
    for b in bs:
        b["out"].fill_(float("nan")); b["dqkv"].fill_(float("nan"))
    torch.cuda.synchronize()
    for s, b in zip(streams, bs):
        with torch.cuda.stream(s):
            if rep % 2:
                (noise @ noise)  # other kernels in the mix
            run(b, s)
    torch.cuda.synchronize()
    for k, b in enumerate(bs):
        for name in ("out", "lse", "dqkv"):
            if not torch.equal(b[name], ref[name]):
                d = (b[name] - ref[name]).abs()
                bad += 1
                print("rep %d stream %d %s differs: max %.3e, %d entries, finite %s" % (rep, k, name, d.max().item(), int((d > 0).sum()),
                                                                                     bool(torch.isfinite(b[name]).all())))
print("mode %d: mismatching results: %d of %d" % (mode, bad, 30 * S * 3))
