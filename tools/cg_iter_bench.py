"""Time of ONE CG iteration (SpMV + update launches) of the label propagation for a batch of E systems, both SpMV forms.
    python tools/cg_iter_bench.py [--workload S] [--episodes 32]"""
import argparse, os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import _lib, ops, synthetic as S  # noqa: E402
from r3dfsseg_amd.batch import EpisodeBatch  # noqa: E402
from r3dfsseg_amd.mpti import MPTI_SelfAtten  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="S")
ap.add_argument("--episodes", type=int, default=32)
ap.add_argument("--train-steps", type=int, default=0, help="optimiser steps on the batch first (the state bench.py measures in)")
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
if args.train_steps:
    from r3dfsseg_amd.dp_train import DPTrainer
    m.train()
    learner = SimpleNamespace(model=m)
    learner.optimizer = torch.optim.Adam(
        [{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
         {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
    learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
    tr = DPTrainer(learner, batch_size=E)
    for _ in range(args.train_steps):
        tr.step([b])
    torch.cuda.synchronize()
    hb = m._head[1]
else:
    with torch.no_grad():
        m.forward_episodes(b)
    hb = m._head[1]
nbr = ops.knn_nodes(hb)
lib = _lib.load()
nnz = sum(int(hb.csr(e)[1][-1].item()) for e in range(E))
nodes = int(hb.desc.view(-1, 32)[:, ops.HD_N_NODES].sum().item())
by = nnz * 6.0 + nodes * (16 + 32 + 32 + 4 * 64) + nodes * (32 + 32 + 32 + 4 * 64)


def solve_ms(iters, reps=4):
    ops.label_propagate(hb, nbr, m.sigma, 0.99, iters, 0.0)
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.label_propagate(hb, nbr, m.sigma, 0.99, iters, 0.0)
    bb.record()
    torch.cuda.synchronize()
    return a.elapsed_time(bb) / reps


for name, mb in (("row-per-wave (r from L2)", 1 << 30), ("128-row workgroups (r in LDS)", 0)):
    old = lib.r3d_debug_set_cg_spmv_lds_min_blocks(mb)
    t = (solve_ms(40) - solve_ms(8)) / 32.0
    lib.r3d_debug_set_cg_spmv_lds_min_blocks(old)
    print("%-32s %7.1f us per iteration (%d systems, nnz %d): %.0f GB/s of algorithmic bytes = %.3f of 8 TB/s" % (
        name, t * 1e3, E, nnz, by / (t * 1e-3) / 1e9, by / (t * 1e-3) / 1e9 / 8000.0), flush=True)
# row lengths of the symmetrised graph (a long row is one wave's work in either SpMV form)
rl = torch.cat([(hb.csr(e)[1][1:] - hb.csr(e)[1][:-1]).float() for e in range(E)])
print("row lengths: mean %.0f  p50 %.0f  p99 %.0f  p99.9 %.0f  max %.0f   (rows %d); rows longer than 1024: %d" % (
    rl.mean(), rl.quantile(0.5), rl.quantile(0.99), rl.quantile(0.999), rl.max(), rl.numel(), int((rl > 1024).sum())))
