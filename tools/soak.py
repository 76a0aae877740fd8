"""A long run of the headline schedule (32 episodes per step, one captured hipGraph per step): episodes/s and the CG
iteration counts of the label propagation window by window -- mpti_train_noise.py:182 trains for 40 000 episodes =
1 250 steps of 32.  usage (GPU box): python tools/soak.py [steps=1250] [window=125] > profiles/rNN_soak.json"""
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import synthetic as S
from r3dfsseg_amd.batch import EpisodeBatch
from r3dfsseg_amd.dp_train import DPTrainer
from r3dfsseg_amd.mpti import MPTI_SelfAtten

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
window = int(sys.argv[2]) if len(sys.argv) > 2 else 125
E = 32
cfg = S.workload_cfg("S")
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
pool = []
for e in range(4 * E):  # four batches of distinct episodes, visited in turn
    data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
    pool.append([t.cuda() for t in data])
batches = [EpisodeBatch.from_episodes(pool[i:i + E]) for i in range(0, len(pool), E)]
learner = SimpleNamespace(model=m)
learner.optimizer = torch.optim.Adam(
    [{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
     {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner, batch_size=E, batch_graph=True)
rows = []
its, mx, n = 0, 0, 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = tr.step([batches[i % len(batches)]])
    its += tr.last_status[2]; mx = max(mx, tr.last_status[3]); n += 1
    if (i + 1) % window == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rows.append(dict(steps="%d-%d" % (i + 1 - window, i + 1), episodes_per_sec=round(window * E / (t1 - t0), 1),
                         ms_per_step=round((t1 - t0) / window * 1e3, 2), cg_iterations_mean=round(its / (n * E), 2),
                         cg_iterations_max=mx, steps_redone_so_far=tr.n_redone, loss=round(float(loss), 4)))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
        its, mx, n = 0, 0, 0
        t0 = time.perf_counter()
print(json.dumps(dict(what="soak of the headline schedule: S3DIS 2-way 5-shot 2048 pts, 32 episodes per step, one MI355X, "
                           "captured hipGraph per step; four batches of distinct synthetic episodes visited in turn",
                      steps=steps, episodes=steps * E, windows=rows), indent=1))
