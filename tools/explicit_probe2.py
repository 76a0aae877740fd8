import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import synthetic as S, train_ops as T, contrast
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.head_train import HeadLPFn
cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
m.att_learner.dropout.p = 0.0
data, _ = S.make_episode(cfg, seed=40, noise_ratio=0.2, train=True)
ep = [t.cuda() for t in data]
sx = ep[0].reshape(4, 9, 512)
# autograd
sfeat = T.get_features_train(m, sx, 2); qfeat = T.get_features_train(m, ep[2], 3)
sfeat.retain_grad(); qfeat.retain_grad()
closs = contrast.per_way_contrast_loss(m, sfeat, ep[1], ep[10])
lploss = HeadLPFn.apply(sfeat, qfeat, m, ep[1], ep[3])
(lploss + 0.1 * closs).backward()
ga = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
# explicit
with torch.no_grad():
    params = T.encoder_params(m)
    cs, cq, cc, ch = (SimpleNamespace(param_list=params) for _ in range(4))
    sf2 = T.EncoderTrainFn.forward(cs, sx, m, 2); qf2 = T.EncoderTrainFn.forward(cq, ep[2], m, 3)
    print("feat equal", torch.equal(sf2, sfeat), torch.equal(qf2, qfeat))
    cl2 = contrast.ContrastFn.forward(cc, sf2, m.proj.weight, m.proj.bias, m, ep[1], ep[10])
    lp2 = HeadLPFn.forward(ch, sf2, qf2, m, ep[1], ep[3])
    one = torch.ones((), device="cuda")
    dsf_c = contrast.ContrastFn.backward(cc, one * 0.1)[0]
    dsf, dqf = HeadLPFn.backward(ch, one)[:2]
    dsf.add_(dsf_c)
    print("dsf diff", (dsf - sfeat.grad).abs().max().item(), sfeat.grad.abs().max().item())
    print("dqf diff", (dqf - qfeat.grad).abs().max().item(), qfeat.grad.abs().max().item())
    gs = T.EncoderTrainFn.backward(cs, sfeat.grad.clone())[3:]
    gq = T.EncoderTrainFn.backward(cq, qfeat.grad.clone())[3:]
    names = {id(p): n for n, p in m.named_parameters()}
    for p, a, b in zip(params, gs, gq):
        if a is None and b is None: continue
        g = (a if a is not None else 0) + (b if b is not None else 0)
        ref = ga[names[id(p)]]
        d = (g.reshape(ref.shape) - ref).abs().max().item()
        if d > 1e-3 * ref.abs().max().item(): print(names[id(p)], d, ref.abs().max().item())
