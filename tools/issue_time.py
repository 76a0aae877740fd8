"""Host-issue time vs GPU time of one step (is the step launch-bound on the Python side?)."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
model = MPTI_SelfAtten(SimpleNamespace(**cfg)); model.load_state_dict(S.make_state_dict(cfg, 123)); model.to(dev)
pool = []
for e in range(4):
    data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
    pool.append([t.to(dev) for t in data])
learner = SimpleNamespace(model=model)
learner.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner)
def run(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
model.train()
print("train  issue %.2f ms  total %.2f ms" % run(lambda i: tr.step([pool[i % 4]])))
model.eval()
def ev(i):
    with torch.no_grad(): model(*pool[i % 4][:4])
print("eval   issue %.2f ms  total %.2f ms" % run(ev))
if os.environ.get("R3D_PROFILE"):
    import cProfile, pstats
    model.train()
    pr = cProfile.Profile(); pr.enable()
    for i in range(20): tr.step([pool[i % 4]])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
