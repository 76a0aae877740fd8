"""Does the head backward depend on where its inputs / outputs live?  Same features, separate tensors vs two halves of
one buffer; and the same call twice (determinism)."""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from r3dfsseg_amd import synthetic as S, train_ops as T, head_train as H
from r3dfsseg_amd.mpti import MPTI_SelfAtten

cfg = S.workload_cfg("S")
data, _ = S.make_episode(cfg, seed=5, noise_ratio=0.2, train=True)
ep = [t.cuda() for t in data]
m = MPTI_SelfAtten(SimpleNamespace(**cfg))
m.load_state_dict(S.make_state_dict(cfg, 123))
m.cuda().train()
m.att_learner.dropout.p = 0.0
Sn, N = cfg["n_way"] * cfg["k_shot"], cfg["pc_npts"]
with torch.no_grad():
    pass
    c = SimpleNamespace(param_list=T.encoder_params(m))
    sf = T.EncoderTrainFn.forward(c, ep[0].reshape(Sn, -1, N), m, 0).clone()
    c = SimpleNamespace(param_list=T.encoder_params(m))
    qf = T.EncoderTrainFn.forward(c, ep[2], m, 0).clone()
    m._lp_force = True

    def run(sfeat, qfeat):
        ch = SimpleNamespace()
        loss = H.HeadLPFn.forward(ch, sfeat, qfeat, m, ep[1], ep[3])
        dsf, dqf = H.HeadLPFn.backward(ch, torch.ones((), device="cuda"))[:2]
        torch.cuda.synchronize()
        return loss.item(), dsf.clone(), dqf.clone(), m._head_buffers(qf.shape[0] // N, sf.device).stats_bwd.clone()

    a = run(sf, qf)
    b = run(sf, qf)
    joint = torch.cat((sf, qf), 0)
    c2 = run(joint[:Sn * N], joint[Sn * N:])
    qf2 = qf + 1e-6 * torch.randn_like(qf)
    d = run(sf, qf2)
for name, r in (("same call again", b), ("inputs as halves of one buffer", c2), ("qfeat + 1e-6 noise", d)):
    print("%-32s loss %.7f vs %.7f  dsf diff %.3e (max %.3e)  dqf diff %.3e (max %.3e)  adjoint stats %s" % (
        name, r[0], a[0], (r[1] - a[1]).abs().max().item(), a[1].abs().max().item(),
        (r[2] - a[2]).abs().max().item(), a[2].abs().max().item(), r[3].tolist()))
