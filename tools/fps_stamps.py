"""Residency timeline of the persistent FPS kernel (library built with -DFPS_STAMPS): per workgroup its start / end time
and the XCC / CU it ran on.  usage (GPU box): python tools/fps_stamps.py [episodes per launch, default 6]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "r3dfsseg_amd", "csrc")
from r3dfsseg_amd import build as B
objs = []
for f in B.SOURCES:
    o = "/tmp/fs_%s.o" % f.replace(".hip", "")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DFPS_STAMPS", "-c", os.path.join(src, f), "-o", o],
                          stderr=subprocess.DEVNULL)
    objs.append(o)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libfps.so"] + objs)
os.environ["R3D_LIB"] = "/tmp/libfps.so"
from types import SimpleNamespace
import numpy as np
import torch
from r3dfsseg_amd import _lib, ops, synthetic as S
from r3dfsseg_amd.batch import EpisodeBatch
from r3dfsseg_amd.mpti import MPTI_SelfAtten
group = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
E = group
eps = []
for e in range(E):
    data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
    eps.append([t.to(dev) for t in data])
b = EpisodeBatch.from_episodes(eps)
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.to(dev).eval()
Sn, N = 10, 2048
with torch.no_grad():
    feat = m.getFeatures_pm(b.x_all.view(E * 12, -1, N), group=12)
hb = m._head_buffers(2, dev, E)
hb.fps_slots = 100000
sy = b.support_y.reshape(E, Sn, N).contiguous()
for it in range(2):
    ops.head_prototypes(hb, sy, None, feat, feat[Sn * N:], 12 * N)
torch.cuda.synchronize()
cdll = _lib.load()._cdll
TB = (10 * 2048 + 255) // 256 + 3
n = TB * E
out = (ctypes.c_ulonglong * (4 * n))()
assert cdll.r3d_fps_debug_read(out, 4 * n) == 0
a = np.array(list(out), dtype=np.uint64).reshape(n, 4)
act = a[:, 3] == 1
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
xcc_ = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
for x_ in range(16):  # the XCDs' s_memtime counters have different origins: times relative to each XCD's first start
    sel_ = xcc_ == x_
    if sel_.any():
        o_ = t0[sel_].min(); t0[sel_] -= o_; t1[sel_] -= o_
base = 0
hw = (a[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
loc = xcc * 1000 + se * 100 + sh * 16 + cu
print("workgroups %d, active %d; span %.1f us (100 MHz ticks)" % (n, act.sum(), (t1.max() - base) / 100.0))
print("active: start min/median/max %.1f / %.1f / %.1f us; duration min/median/max %.1f / %.1f / %.1f us" % (
    (t0[act].min() - base) / 100.0, (np.median(t0[act]) - base) / 100.0, (t0[act].max() - base) / 100.0,
    (t1[act] - t0[act]).min() / 100.0, np.median(t1[act] - t0[act]) / 100.0, (t1[act] - t0[act]).max() / 100.0))
late = act & (t0 - base > 5000)
print("active workgroups that started more than 50 us after the first: %d" % late.sum())
locs, cnt = np.unique(loc[act], return_counts=True)
print("distinct (xcc, se, sh, cu) locations used by active workgroups: %d; workgroups per location: max %d, histogram %s" % (
    len(locs), cnt.max(), np.bincount(cnt).tolist()))
print("active workgroups per XCC:", np.bincount(xcc[act], minlength=8).tolist())
for e in range(E):
    sl = slice(TB * e, TB * (e + 1))
    print("episode %d: active %d, start %.1f .. %.1f us, end %.1f .. %.1f us" % (
        e, act[sl].sum(), (t0[sl][act[sl]].min() - base) / 100.0, (t0[sl][act[sl]].max() - base) / 100.0,
        (t1[sl][act[sl]].min() - base) / 100.0, (t1[sl][act[sl]].max() - base) / 100.0))
