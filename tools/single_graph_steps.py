"""A few single-episode training steps as ONE captured graph (the reference's schedule: one episode per Adam step), for a
rocprofv3 kernel trace: where the 9 ms of a replay go.   usage: single_graph_steps.py [steps=20]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = S.workload_cfg("S")
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
eps = [[t.cuda() for t in S.make_episode(cfg, seed=1000 + e, noise_ratio=0.2, train=True)[0]] for e in range(4)]
learner = SimpleNamespace(model=m)
learner.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner, n_slots=1, example=eps[0])
for i in range(3):
    tr.step([eps[i % 4]])
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    tr.step([eps[i % 4]])
torch.cuda.synchronize()
print("single-episode graph step: %.2f ms  (%.1f episodes/s), redone %d" % ((time.time() - t0) / steps * 1e3, steps / (time.time() - t0), tr.n_redone))
