"""Eager training steps at workload S with finiteness checks after every episode (debugging aid)."""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
model = MPTI_SelfAtten(SimpleNamespace(**cfg)); model.load_state_dict(S.make_state_dict(cfg, 123)); model.to(dev)
learner = SimpleNamespace(model=model)
learner.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner)
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for step in range(n_steps):
    eps = []
    for j in range(4):
        data, _ = S.make_episode(cfg, seed=1000 + 4 * step + j, noise_ratio=0.2, train=True)
        eps.append([t.to(dev) for t in data])
    loss = tr.step(eps)
    torch.cuda.synchronize()
    hb = model._head[1]
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print("step %d loss %.5f grad max %.3e params finite %s  fwd stats %s bwd stats %s" % (
        step, float(loss), gmax, finite, hb.stats.tolist(), hb.stats_bwd.tolist()), flush=True)
    if not finite or not (gmax < 1e6):
        print("STOP: non-finite or exploding"); break
