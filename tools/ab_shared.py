"""A/B of the shared support+query launch sequence against two separate sequences: features, feature gradients and
parameter gradients of one training episode at workload S (run on the GPU box)."""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from r3dfsseg_amd import synthetic as S, train_ops as T
from r3dfsseg_amd.mpti import MPTI_SelfAtten

cfg = S.workload_cfg(sys.argv[1] if len(sys.argv) > 1 else "S")
data, _ = S.make_episode(cfg, seed=5, noise_ratio=0.2, train=True)
ep = [t.cuda() for t in data]
res = {}
for shared in (False, True):
    T.SHARED_LAUNCHES = shared
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(S.make_state_dict(cfg, 123))
    m.cuda().train()
    m.att_learner.dropout.p = 0.0
    m._trace = {}
    out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
            lp_iters=m.lp_max_iter)
    (out[1] + 0.1 * out[2]).backward()
    tr = m._trace
    res[shared] = dict(sfeat=tr["sfeat"].detach().clone(), qfeat=tr["qfeat"].detach().clone(),
                       dsf=tr["sfeat"].grad.clone(), dqf=tr["qfeat"].grad.clone(), lp=out[1].item(), cl=out[2].item(),
                       grads={n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                       rm={n: b.clone() for n, b in m.named_buffers()})
a, b = res[False], res[True]
print("loss", a["lp"], b["lp"], a["cl"], b["cl"])
for k in ("sfeat", "qfeat", "dsf", "dqf"):
    d = (a[k] - b[k]).abs()
    print(k, "max abs diff %.3e  (max |.| %.3e)  rows differing %d" % (d.max().item(), a[k].abs().max().item(),
                                                                   int((d.amax(1) > 0).sum())))
worst = 0
for n in a["grads"]:
    d = (a["grads"][n] - b["grads"][n]).abs().max().item() / max(1e-12, a["grads"][n].abs().max().item())
    worst = max(worst, d)
    if d > 1e-5:
        print("  grad", n, "%.2e" % d)
print("worst parameter-gradient difference (max-rel) %.2e" % worst)
wb = max(((a["rm"][n].float() - b["rm"][n].float()).abs().max().item() for n in a["rm"]), default=0)
print("worst running-statistic difference %.2e" % wb)

# ---- where does a difference in d qfeat come from?  the head alone, on the traced features of the shared run
from r3dfsseg_amd import head_train as H
with torch.no_grad():
    m._lp_force = True
    ch = SimpleNamespace()
    H.HeadLPFn.forward(ch, b["sfeat"], b["qfeat"], m, ep[1], ep[3])
    dsf, dqf = H.HeadLPFn.backward(ch, torch.ones((), device="cuda"))[:2]
    print("head alone on the shared run's features: d qfeat vs the run's own %.3e, vs the separate run's %.3e (max %.3e)" % (
        (dqf - b["dqf"]).abs().max().item(), (dqf - a["dqf"]).abs().max().item(), dqf.abs().max().item()))
    hb = m._head_buffers(ep[2].shape[0], dqf.device)
    print("stats fwd", hb.stats.tolist(), "bwd", hb.stats_bwd.tolist())
