"""Single-episode EAGER training steps (the reference's schedule through DPTrainer._eager_pass): time per step and what
every attempt's status words said (a first attempt that misses is redone on the conservative schedule: twice the time).
usage: single_eager_probe.py [steps=20] [pre_steps=0: batched optimiser steps before, as bench.py's steady state]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.batch import EpisodeBatch
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = S.workload_cfg("S")
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
eps = [[t.cuda() for t in S.make_episode(cfg, seed=1000 + e, noise_ratio=0.2, train=True)[0]] for e in range(32)]
learner = SimpleNamespace(model=m)
learner.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner, batch_size=32)
batch = [EpisodeBatch.from_episodes(eps)]
for i in range(pre):
    tr.step(batch)
log = []
orig = m.lp_converged
def logged(backward=False):
    ok = orig(backward=backward)
    hb = m._head[1]
    log.append((ok, hb.stats.view(-1)[:2].tolist(), hb.knn_status.tolist(), int(hb.desc.view(-1, 32)[0, ops.HD_FPS_TIMEOUT]),
                hb.stats_bwd.view(-1)[:2].tolist(), m._lp_budget))
    return ok
m.lp_converged = logged
saved = tr.runner, tr.graphs
tr.runner, tr.graphs = None, None
for i in range(3):
    tr.step([eps[i % 32]])
torch.cuda.synchronize()
log.clear()
t0 = time.time()
for i in range(steps):
    tr.step([eps[(3 + i) % 32]])
torch.cuda.synchronize()
dt = time.time() - t0
print("single-episode eager step: %.2f ms (%.1f episodes/s); attempts %d for %d steps" % (dt / steps * 1e3, steps / dt, len(log), steps))
for l in log[:12]:
    print("  ok=%s fwd(conv,iters)=%s knn_status=%s fps_timeout=%d bwd(conv,iters)=%s budget=%s" % l)
