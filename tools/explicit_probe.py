import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.head_train import explicit_train_episode
cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
m.att_learner.dropout.p = 0.0
data, _ = S.make_episode(cfg, seed=40, noise_ratio=0.2, train=True)
ep = [t.cuda() for t in data]
out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10])
(out[1] + 0.1 * out[2]).backward()
names = [n for n, p in m.named_parameters() if p.requires_grad]
ps = [p for p in m.parameters() if p.requires_grad]
sink = [torch.zeros_like(p) for p in ps]
loss, _, _ = explicit_train_episode(m, ep, sink)
print("loss", float(out[1] + 0.1 * out[2]), float(loss))
for n, p, g in zip(names, ps, sink):
    a = p.grad
    if a is None:
        print(n, "autograd None, explicit max", g.abs().max().item()); continue
    d = (a - g).abs().max().item(); sc = a.abs().max().item()
    if d > 1e-3 * max(sc, 1e-6): print("%-40s diff %.3e scale %.3e" % (n, d, sc))
