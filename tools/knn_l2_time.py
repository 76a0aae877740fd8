"""Time the head's 201-NN (r3d_knn_topk mode L2, k = 201, n = 4396, C = 192) launch by launch with HIP events.
usage: knn_l2_time.py [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4396
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
x = (torch.randn(n, 192, device="cuda") * 0.15).contiguous()
nv = torch.tensor([n], device="cuda", dtype=torch.int32)
st = torch.zeros(1, device="cuda", dtype=torch.int32)
ts = []
for i in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.knn(x, 1, n, 201, mode=ops.SCORE_L2, n_valid=nv, status=st)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print("201-NN n=%d: first %.3f ms, then" % (n, ts[0]), " ".join("%.3f" % t for t in ts[1:]), "status", int(st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(20):
    ops.knn(x, 1, n, 201, mode=ops.SCORE_L2, n_valid=nv, status=st)
b.record(); torch.cuda.synchronize()
print("20 back to back: %.3f ms each" % (a.elapsed_time(b) / 20))
# clustered features (a trained encoder separates the classes: many near-equal distances inside a cluster)
for spread in (0.05, 0.01, 0.002):
    lab = torch.randint(0, 3, (n,), device="cuda")
    xc = (torch.randn(3, 192, device="cuda") * 0.3)[lab] + torch.randn(n, 192, device="cuda") * spread
    ops.knn(xc.contiguous(), 1, n, 201, mode=ops.SCORE_L2, n_valid=nv, status=st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(10):
        ops.knn(xc.contiguous(), 1, n, 201, mode=ops.SCORE_L2, n_valid=nv, status=st)
    b.record(); torch.cuda.synchronize()
    print("3 clusters, spread %.3f: %.3f ms each, status %d" % (spread, a.elapsed_time(b) / 10, int(st)))
# the bench's warm-cache repeat timer around the same call
timer = ops.KernelTimer(["knn_topk_l2"], repeat=8)
ops.set_timer(timer)
for i in range(5):
    ops.knn(x, 1, n, 201, mode=ops.SCORE_L2, n_valid=nv, status=st)
ops.set_timer(None); timer.close()
print("KernelTimer(repeat=8):", timer.summary())
