"""After N training steps: replay single episodes through slot 0's graph until one does not converge, then run the SAME
episode with the SAME dropout seed eagerly through the same slot state and compare (bx3 and fp32 arithmetic)."""
import sys
from types import SimpleNamespace
import torch
sys.path.insert(0, ".")
from r3dfsseg_amd import _lib, ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
from r3dfsseg_amd.dp_train import DPTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
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
for i in range(steps):
    trainer.step(pool)
torch.cuda.synchronize()
print("trained %d steps, redone %d" % (steps, trainer.n_redone), flush=True)
G = trainer.graphs
default_slot = model._slot
found = 0
for trial in range(300):
    eps = [pool[(6 * trial + k) % 32] for k in range(6)]
    seeds = [int(sl.state.seed_dev.item()) for sl in G.slots]
    G.run(eps, apply_bn=False)
    bad, ovf, its, mx = G.step_status()
    if not bad:
        continue
    c = G.counters.tolist()
    for k, sl in enumerate(G.slots):
        if c[k][0] == 0:
            continue
        hb = sl.state.last[1]
        n = int(hb.desc[ops.HD_N_NODES])
        print("trial %d slot %d: replay bad, fwd %s adjoint %s FPS time-out %d 201-NN %s | nodes finite %s" % (
            trial, k, hb.stats.tolist(), hb.stats_bwd.tolist(), int(hb.desc[ops.HD_FPS_TIMEOUT]), hb.knn_status.tolist(),
            bool(torch.isfinite(hb.nodes[:n]).all())), flush=True)
        # 1. the same solve again, alone on the device, from the node matrix the replay left behind
        Z_replay, stats_replay = hb.Z[:n].clone(), hb.stats.clone()
        n_, rp, col, val = hb.csr()
        rp0, col0, val0 = rp.clone(), col.clone(), val.clone()
        Y0 = hb.Y[:n].clone()
        nbr = ops.knn(hb.nodes, 1, hb.n_cap, hb.kp1, mode=ops.SCORE_L2, n_valid=hb.desc[ops.HD_N_NODES:], status=hb.knn_status)
        ops.label_propagate(hb, nbr, model.sigma, 0.99, 200, model.lp_tol)
        torch.cuda.synchronize()
        print("   same node matrix solved again alone: %s ; Z vs the replay's: max diff %.3e" % (
            hb.stats.tolist(), (hb.Z[:n] - Z_replay).abs().max().item()), flush=True)
        n_, rp1, col1, val1 = hb.csr()
        nnz0, nnz1 = int(rp0[n]), int(rp1[n])
        same_rp = torch.equal(rp0[:n + 1], rp1[:n + 1])
        print("   CSR of the replay vs rebuilt: nnz %d / %d, row_ptr equal %s, col equal %s, val max diff %.3e, Y equal %s" % (
            nnz0, nnz1, same_rp, same_rp and torch.equal(col0[:nnz0], col1[:nnz0]),
            (val0[:min(nnz0, nnz1)] - val1[:min(nnz0, nnz1)]).abs().max().item() if same_rp else float("nan"),
            torch.equal(Y0, hb.Y[:n])), flush=True)
        # 2. the same episode with the same dropout seed, eagerly through the same slot state
        nodes_replay = hb.nodes[:n].clone()
        for mode in (1, 0):
            _lib.check(lib.r3d_set_matrix_arith(mode))
            sl.state.seed_dev.fill_(seeds[k])
            model._slot = sl.state
            saved_budget = sl.state.fixed_budget
            sl.state.fixed_budget = None
            ep = eps[k]
            out = model(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
                        lp_iters=model.lp_max_iter)
            torch.cuda.synchronize()
            hb2 = model._slot.last[1]
            d = (hb2.nodes[:n] - nodes_replay).abs()
            print("   eager arith %d, same seed: fwd %s | nodes vs replay: max diff %.3e (max |.| %.3g), rows differing > 1e-4: %d" % (
                mode, hb2.stats.tolist(), d.max().item(), nodes_replay.abs().max().item(), int((d.amax(1) > 1e-4).sum())), flush=True)
            sl.state.fixed_budget = saved_budget
        _lib.check(lib.r3d_set_matrix_arith(1))
        model._slot = default_slot
        found += 1
    if found >= 3:
        break
print("bad replays examined:", found)
