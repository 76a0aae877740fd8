"""A few episode-batched steps of a workload (the headline schedule of bench.py: E episodes per launch sequence) with a
small launch count, for rocprofv3 kernel traces and PMC passes.  Ends with two calibration launches over 1 GiB (beyond
the 256 MiB Infinity Cache): a 4-byte-per-lane streaming copy (r3d_copy_cols_kernel) and a 16-byte-per-lane one
(r3d_fill... no: torch's vectorised copy), whose FETCH_SIZE / WRITE_SIZE readings against the known byte counts give
the counter factors of MI355X_MICROARCH.md's HBM section for OUR access widths.
usage: one_step.py [train|eval] [S|C] [episodes=32] [steps=3]"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.batch import EpisodeBatch
from r3dfsseg_amd.mpti import MPTI_SelfAtten
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
W = sys.argv[2] if len(sys.argv) > 2 else "S"
E = int(sys.argv[3]) if len(sys.argv) > 3 else 32
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cfg = S.workload_cfg(W)
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train(mode == "train")
eps = []
for e in range(E):
    data, _ = S.make_episode(cfg, seed=1000 + e, noise_ratio=0.2, train=True)
    eps.append([t.cuda() for t in data])
batch = [EpisodeBatch.from_episodes(eps)]
if mode == "train":
    from r3dfsseg_amd.dp_train import DPTrainer
    learner = SimpleNamespace(model=m)
    learner.optimizer = torch.optim.Adam(
        [{'params': m.encoder.parameters(), 'lr': 0.0001}, {'params': m.base_learner.parameters()},
         {'params': m.att_learner.parameters()}, {'params': m.proj.parameters()}], lr=1e-3)
    learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
    tr = DPTrainer(learner, batch_size=E)
    for it in range(steps):
        tr.step(batch)
        torch.cuda.synchronize()
        print("step", it, "status", tr.last_status, "redone", tr.n_redone, flush=True)
else:
    from r3dfsseg_amd.batched import EpisodeBatchRunner
    run = EpisodeBatchRunner(m)
    for it in range(steps):
        run.begin_step()
        run.eval_batch(batch[0])
        print("step", it, "status", run.step_status(), flush=True)
# calibration: known byte counts, 1 GiB each way
n = 1 << 28
a = torch.empty(n // 64, 64, device="cuda", dtype=torch.float32).normal_()
b = torch.empty_like(a)
ops.copy_cols(a, b)            # r3d_copy_cols_kernel: 4 B per lane, coalesced
c = a.clone()                  # at::native vectorised copy: 16 B per lane
torch.cuda.synchronize()
print("calibration launches done: %d bytes read and written by each" % (4 * n), flush=True)
