"""Train workload S for a number of steps (captured episodes), then run one eager training forward and dump the
label-propagation system of that episode -- CSR of S, right-hand side, solution, CG iterations -- to an .npz for
offline solver experiments.  usage: dump_lp_system.py STEPS OUT.npz"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer
steps, out = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
model = MPTI_SelfAtten(SimpleNamespace(**cfg)); model.load_state_dict(S.make_state_dict(cfg, 123)); model.to(dev)
pool = []
for e in range(64):
    data, _ = S.make_episode(cfg, seed=1000 + e, noise_ratio=0.2, train=True)
    pool.append([t.to(dev) for t in data])
learner = SimpleNamespace(model=model)
learner.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
tr = DPTrainer(learner, n_slots=6, example=pool[0])
for i in range(steps):
    tr.step([pool[(32 * i + j) % 64] for j in range(32)])
    if i % 20 == 0:
        torch.cuda.synchronize(); print("step", i, flush=True)
torch.cuda.synchronize()
print("graphs check", tr.graphs.check())
model.eval()
with torch.no_grad():
    model(*pool[3][:4], lp_iters=model.lp_max_iter)
torch.cuda.synchronize()
hb = model._head[1]
n, row_ptr, col, val = hb.csr()
row_ptr, col, val = row_ptr.cpu().numpy(), col.cpu().numpy().astype(np.int32), val.cpu().numpy()
nnz = int(row_ptr[n])
np.savez_compressed(out, n=n, row_ptr=row_ptr[:n + 1].copy(), col=col[:nnz].copy(), val=val[:nnz].copy(),
                    Y=hb.Y.cpu().numpy()[:n], Z=hb.Z.cpu().numpy()[:n], stats=hb.stats.cpu().numpy(),
                    nodes=hb.nodes.cpu().numpy()[:n].astype(np.float16), n_proto=int(hb.desc[ops.HD_N_PROTO].item()),
                    query_y=pool[3][3].cpu().numpy())
print("dumped n", n, "nnz", nnz, "stats (converged, iterations)", hb.stats.cpu().numpy())
