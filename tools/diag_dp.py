import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from types import SimpleNamespace
from r3dfsseg_amd import synthetic as S
import test_gpu_batched as TB
from r3dfsseg_amd.dp_train import DPTrainer
cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
eps = TB._episodes(cfg, 3)
res = {}
for mode in ("eager", "batched"):
    m = TB._model(cfg, True, 0.1)
    learner = SimpleNamespace(model=m)
    learner.optimizer = torch.optim.Adam([{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
             {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
    learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
    tr = DPTrainer(learner, batch_size=2 if mode == "batched" else 0)
    l1 = float(tr.step(eps))
    g = {n: p.grad.clone() if p.grad is not None else None for n, p in m.named_parameters()}
    l2 = float(tr.step(eps))
    res[mode] = (l1, l2, {n: p.detach().clone() for n, p in m.named_parameters()}, g)
a, b = res["eager"], res["batched"]
print(a[0], b[0], a[1], b[1])
for n in a[2]:
    d = (a[2][n] - b[2][n]).abs().max().item()
    ga, gb = a[3][n], b[3][n]
    gd = (ga - gb).abs().max().item() if ga is not None else -1
    gm = ga.abs().max().item() if ga is not None else -1
    if d > 1e-6: print("%-50s dparam %.2e  dgrad %.2e  |grad| %.2e" % (n, d, gd, gm))
