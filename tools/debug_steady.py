"""Train like bench.py's steady-state leg and find out why label propagation stops converging (if it does):
usage (GPU box): python tools/debug_steady.py [steps]"""
import sys
from types import SimpleNamespace
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = S.workload_cfg("S")
model = MPTI_SelfAtten(SimpleNamespace(**cfg))
model.load_state_dict(S.make_state_dict(cfg, 123))
model.to(dev).train()
pool = []
for e in range(32):
    data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
    pool.append([t.to(dev) for t in data])
learner = SimpleNamespace(model=model)
learner.optimizer = torch.optim.Adam(
    [{'params': model.encoder.parameters(), 'lr': 0.0001}, {'params': model.base_learner.parameters()},
     {'params': model.att_learner.parameters()}, {'params': model.proj.parameters()}], lr=1e-3)
learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
trainer = DPTrainer(learner, n_slots=6, example=pool[0])
first_bad = None
import os
stop_at_first = os.environ.get("STOP_AT_FIRST", "1") == "1"
last_redone = 0
for i in range(steps):
    loss = trainer.step(pool)
    if i % 10 == 9 or trainer.n_redone > last_redone:
        c = trainer.graphs.counters
        print("step %3d loss %.4f redone %d  last run: bad %d max its %d  enabled budget %d" % (
            i, float(loss), trainer.n_redone, int(c[:, 0].sum()), int(c[:, 3].max()), trainer.graphs.active_budget), flush=True)
    last_redone = trainer.n_redone
    if not stop_at_first:
        continue
    if trainer.n_redone and first_bad is None:
        first_bad = i
        from r3dfsseg_amd import ops
        for k, sl in enumerate(trainer.graphs.slots):
            hb = sl.state.last[1]
            n = int(hb.desc[ops.HD_N_NODES])
            print("  slot %d: fwd (converged, its) %s  adjoint %s  FPS time-out %d  201-NN status %s  n_nodes %d n_proto %d  nodes finite %s max %.3g  Z finite %s" % (
                k, hb.stats.tolist(), hb.stats_bwd.tolist(), int(hb.desc[ops.HD_FPS_TIMEOUT]), hb.knn_status.tolist(), n,
                int(hb.desc[ops.HD_N_PROTO]), bool(torch.isfinite(hb.nodes[:n]).all()), hb.nodes[:n].abs().max().item(),
                bool(torch.isfinite(hb.Z[:n]).all())), flush=True)
        print("  counters of that run:", trainer.graphs.counters.tolist(), "active budget", trainer.graphs.active_budget)
        break
torch.cuda.synchronize()
print("first redone step:", first_bad)
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
print("non-finite parameters:", bad)
print("att temperature", model.att_learner.temperature, "max |w|:", {n: round(p.abs().max().item(), 3) for n, p in model.named_parameters() if "att" in n})
# one eager episode per arithmetic on the current weights: does the forward LP converge, are the features finite?
model.att_learner.dropout.p = 0.0
for mode in (1, 0):
    _lib.check(lib.r3d_set_matrix_arith(mode))
    rows = []
    for e in range(8):
        ep = pool[e]
        model._trace = {}
        out = model(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
                    lp_iters=model.lp_max_iter)
        sf, qf = model._trace["sfeat"], model._trace["qfeat"]
        hb = model._head_buffers(ep[2].shape[0], dev)
        rows.append((bool(torch.isfinite(sf).all() and torch.isfinite(qf).all()), sf.abs().max().item(), qf[:, 64:128].abs().max().item(),
                     hb.stats.tolist(), float(out[1])))
    print("arith %d:" % mode)
    for r in rows:
        print("   features finite %s  max |sfeat| %.3g  max |att part of qfeat| %.3g  LP (converged, iterations) %s  lp_loss %.4f" % r)
_lib.check(lib.r3d_set_matrix_arith(1))
