"""Train workload S for STEPS optimiser steps on the headline schedule (32 episodes per step, batched), then dump the
label-propagation systems of a few episodes of the LAST step -- CSR of S, right-hand side, solution, CG
iterations, node features (fp16), prototype count -- one .npz per episode, for offline solver experiments
(tools/lp_coarse_study.py).  usage: dump_lp_systems.py STEPS OUT_PREFIX [episode ...]"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
steps, out = int(sys.argv[1]), sys.argv[2]
which = [int(a) for a in sys.argv[3:]] or [0, 5, 13, 22, 31]
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
model = MPTI_SelfAtten(SimpleNamespace(**cfg)); model.load_state_dict(S.make_state_dict(cfg, 123)); model.to(dev)
pool = []
for e in range(128):
    data, _ = S.make_episode(cfg, seed=1000 + e, noise_ratio=0.2, train=True)
    pool.append([t.to(dev) for t in data])
learner = SimpleNamespace(model=model)
learner.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner, batch_size=32)
for i in range(steps):
    tr.step([pool[(32 * i + j) % 128] for j in range(32)])
    if i % 100 == 0:
        torch.cuda.synchronize(); print("step", i, "redone", tr.steps_redone if hasattr(tr, "steps_redone") else "-", flush=True)
torch.cuda.synchronize()
hb = model._head[1]  # the head buffers of the last training step's 32 episodes (training-mode features, dropout on)
E, cap = hb.E, hb.n_cap
stats = hb.stats.view(E, 2).cpu().numpy()
print("last step: CG iterations per system", stats[:, 1].tolist(), flush=True)
for k in which:
    e = k % E
    n, row_ptr, col, val = hb.csr(e)
    row_ptr, col, val = row_ptr.cpu().numpy(), col.cpu().numpy().astype(np.int32), val.cpu().numpy()
    nnz = int(row_ptr[n])
    sl = slice(e * cap, e * cap + n)
    o = hb.lp_off["dinv"]
    dinv = hb.lp_ws.view(-1, hb.lp_stride)[e][o:o + n].view(torch.float32).cpu().numpy()
    np.savez_compressed("%s_%d_ep%d.npz" % (out, steps, e), n=n, row_ptr=row_ptr[:n + 1].copy(), col=col[:nnz].copy(),
                        val=val[:nnz].copy(), Y=hb.Y.cpu().numpy()[sl], Z=hb.Z.cpu().numpy()[sl], stats=stats[e], dinv=dinv,
                        nodes=hb.nodes.cpu().numpy()[sl].astype(np.float16),
                        n_proto=int(hb.desc.view(E, 32)[e, ops.HD_N_PROTO].item()))
    print("episode", e, "n", n, "nnz", nnz, "stats (converged, iterations)", stats[e], flush=True)
