"""Phase stamps of r3d_cg_update_kernel (iteration 3, workgroup 0 and the last-arriving workgroup), from a library built
with -DCG_STAMPS.  usage (GPU box): python tools/cg_stamps.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "r3dfsseg_amd", "csrc")
objs = []
for f in ("error", "knn", "gemm", "train_ops", "head_graph"):
    o = "/tmp/st_%s.o" % f
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
                           "-DCG_STAMPS", "-c", os.path.join(src, f + ".hip"), "-o", o])
    objs.append(o)
import torch
lib = ctypes.CDLL("/dev/null") if False else None
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libst.so"] + objs)
L = ctypes.CDLL("/tmp/libst.so")
n, kp1, D = 4396, 201, 192
x = (torch.randn(n, D, device="cuda") * 0.12).contiguous()
Y = torch.zeros(n, 4, device="cuda"); Y[torch.arange(300), torch.randint(0, 3, (300,))] = 1
Z = torch.empty(n, 4, device="cuda")
nd = torch.tensor([n], device="cuda", dtype=torch.int32); npd = torch.tensor([300], device="cuda", dtype=torch.int32)
p = lambda t: ctypes.c_void_p(t.data_ptr())
L.r3d_knn_norm_ws_words.restype = ctypes.c_long; L.r3d_lp_ws_words.restype = ctypes.c_long; L.r3d_cm_pitch.restype = ctypes.c_long
norm = torch.empty(L.r3d_knn_norm_ws_words(1, n), device="cuda"); cm = torch.empty(D * L.r3d_cm_pitch(n), device="cuda")
nbr = torch.empty(n, kp1, device="cuda", dtype=torch.int32); st = torch.zeros(1, device="cuda", dtype=torch.int32)
L.r3d_knn_topk(p(x), ctypes.c_long(D), None, 1, n, D, kp1, 1, p(nd), p(norm), p(cm), p(nbr), None, p(st), None)
ws = torch.empty(L.r3d_lp_ws_words(n, kp1), device="cuda", dtype=torch.int32); stats = torch.zeros(2, device="cuda", dtype=torch.int32)
for rep in range(3):
    rc = L.r3d_label_propagate(p(x), ctypes.c_long(D), D, p(nbr), kp1, p(Y), p(nd), p(npd), n, ctypes.c_float(1.0), ctypes.c_float(0.99), 12,
                               ctypes.c_float(0.0), p(Z), p(ws), ctypes.c_long(ws.numel()), p(stats), None)
    assert rc == 0
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    assert L.r3d_cg_debug_read(out) == 0
    t = list(out)
    print("wg0: loads+pq reduce %d | update %d | partials %d | drain+ticket %d   last wg: reduce step %d   (s_memtime ticks, 100 MHz: x10 ns)" % (
        t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[17] - t[16]))
