"""Episode-batched execution: the E episodes of a batch go through ONE launch sequence (round 3).

Why: one episode is ~900 launches, most of them latency chains -- 100 dependent FPS rounds, two launches per CG
iteration and 15-30 iterations per solve, reductions of a few workgroups -- and a single system's kernels cannot fill
256 CUs (the 201-NN of one graph is 138 workgroups).  Episodes are independent units (the reference runs
them one per step, mpti_train_noise.py:57-98), so the MI355X-native schedule batches them INSIDE the kernels:
kNN / GEMMs / EdgeConv / attention see E (S + Q) clouds in one grid; BatchNorm keeps per-(episode, call) statistics
(segments, train_ops.py); the FPS launch carries the segments of as many episodes as fit the chip together; every CG
iteration is two launches for all E systems (each with its own convergence flag), whose SpMV streams E matrices and
is genuinely HBM bound; weight gradients are summed over the batch inside the dW GEMMs.  Per-episode results equal the
one-episode path (tests/test_gpu_batched.py).  Round 2 instead replayed one hipGraph per episode on 6 streams, hiding
the chains behind each other; that path (episode_graph.py) remains for the single-episode learner.

EpisodeBatchRunner owns what a step needs around head_train.explicit_train_batch: the deferred BatchNorm
running-statistics records (applied only when the step is kept), the per-step solver status counters, and the eval
forward over batches."""
import torch

from . import ops, train_ops as T
from .batch import EpisodeBatch
from .head_train import explicit_train_batch


class EpisodeBatchRunner:
    def __init__(self, model, max_episodes=256):
        self.model = model
        dev = next(model.parameters()).device
        self.dev = dev
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.max_episodes = max_episodes
        self.bn_records = T.BNRecorder(max_episodes, dev)
        self.rec_index = torch.zeros(1, device=dev, dtype=torch.int32)  # first record of the running batch: 2 * e0
        self.bn_records.index_dev = self.rec_index
        # per step: [systems not converged (forward + adjoint) + FPS time-outs, 201-NN overflows, CG iterations (sum), (max)]
        self.counters = torch.zeros(4, device=dev, dtype=torch.int64)
        self._pinned = torch.zeros(4, dtype=torch.int64).pin_memory()
        self.n_done = 0  # episodes of the running step

    # ------------------------------------------------------------------ status
    def _count(self, backward):
        hb = self.model._head[1]
        E = hb.E
        c = self.counters
        st = hb.stats.view(E, 2)
        bad = (st[:, 0] == 0).sum() + (hb.desc.view(E, 32)[:, ops.HD_FPS_TIMEOUT] != 0).sum()
        if backward:
            bad = bad + (hb.stats_bwd.view(E, 2)[:, 0] == 0).sum()
        c[0] += bad
        c[1] += hb.knn_status[0].clamp(max=1)
        c[2] += st[:, 1].sum()
        torch.maximum(c[3], st[:, 1].max().to(torch.int64), out=c[3])

    def begin_step(self):
        self.counters.zero_()
        self.n_done = 0

    def step_status(self):
        """Host wait: (systems that did not converge / timed out, batches with a 201-NN overflow, CG iterations sum, max)
        of the batches run since begin_step()."""
        self._pinned.copy_(self.counters, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return tuple(int(v) for v in self._pinned)

    # ------------------------------------------------------------------ training
    def train_batch(self, batch, grad_sink, loss_weight=0.1):
        """Forward + backward of one EpisodeBatch into grad_sink (accumulating).  BatchNorm batch statistics are RECORDED
        (records 2 (n_done + e) + p); apply_running_stats() folds them into the running statistics once the step is kept.
        Returns (loss (E,), logits, metrics (E, 4), lp_loss (E,), contrast_loss (E,))."""
        assert self.n_done + batch.E <= self.max_episodes
        m = self.model
        m._lp_force = False
        self.rec_index.fill_(2 * self.n_done)
        saved = T.bn_recorder
        T.bn_recorder = self.bn_records
        try:
            out = explicit_train_batch(m, batch, grad_sink, loss_weight)
        finally:
            T.bn_recorder = saved
        self._count(backward=True)
        self.n_done += batch.E
        return out

    def apply_running_stats(self):
        """The running statistics after the step's episodes, in episode order (support call, then query call each)."""
        if self.n_done:
            self.bn_records.apply(self.n_done)

    # ------------------------------------------------------------------ inference
    def eval_batch(self, batch, eval=False):
        """(logits (E, n_q, n_way + 1, N), loss (E,)) of one EpisodeBatch; status counted like the training batches."""
        with torch.no_grad():
            out = self.model.forward_episodes(batch, eval=eval)
        self._count(backward=False)
        self.n_done += batch.E
        return out


def collate(episodes, batch_size):
    """Episode lists (train or test layout) -> EpisodeBatch objects of up to batch_size episodes."""
    return [EpisodeBatch.from_episodes(episodes[i:i + batch_size]) for i in range(0, len(episodes), batch_size)]
