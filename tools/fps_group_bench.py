"""Time r3d_head_prototypes_batched on E episodes of a workload for several FPS group sizes (episodes per persistent
launch).  python tools/fps_group_bench.py [--workload S] [--episodes 32]"""
import argparse, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import ops, synthetic as S  # noqa: E402
from r3dfsseg_amd.batch import EpisodeBatch  # noqa: E402
from r3dfsseg_amd.mpti import MPTI_SelfAtten  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="S")
ap.add_argument("--episodes", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = S.workload_cfg(args.workload)
E = args.episodes
eps = []
for e in range(E):
    data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
    eps.append([t.to(dev) for t in data])
b = EpisodeBatch.from_episodes(eps)
m = MPTI_SelfAtten(SimpleNamespace(**cfg))
m.load_state_dict(S.make_state_dict(cfg, 123))
m.to(dev).eval()
Sn, N = cfg["n_way"] * cfg["k_shot"], cfg["pc_npts"]
n_q = b.query_x.shape[1]
with torch.no_grad():
    feat = m.getFeatures_pm(b.x_all.view(E * (Sn + n_q), -1, N), group=Sn + n_q)
hb = m._head_buffers(n_q, dev, E)
sy = b.support_y.reshape(E, Sn, N).contiguous()
ep_rows = (Sn + n_q) * N
ref = None
for slots in (1000, 750, 670, 590, 500, 340, 250, 90):
    hb.fps_slots = slots
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.head_prototypes(hb, sy, None, feat, feat[Sn * N:], ep_rows)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 2
    nodes = hb.nodes.clone()
    same = ref is None or torch.equal(nodes, ref)
    ref = nodes if ref is None else ref
    to = int(hb.desc.view(E, 32)[:, ops.HD_FPS_TIMEOUT].sum())
    print("fps_slots %3d  group %d  launches %2d  head_prototypes %.2f ms  (%.3f ms/episode)  same %s timeouts %d" % (
        slots, hb.fps_group, -(-E // hb.fps_group), el * 1e3, el * 1e3 / E, same, to), flush=True)
hb.fps_one_launch = False
for it in range(2):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.head_prototypes(hb, sy, None, feat, feat[Sn * N:], ep_rows)
torch.cuda.synchronize()
print("one launch per round: %.2f ms  same %s" % ((time.perf_counter() - t0) * 1e3, torch.equal(hb.nodes, ref)))
