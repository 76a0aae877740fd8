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
    def train_batch_graph(self, batch, grad_sink, loss_weight=0.1):
        """train_batch through a captured hipGraph of the batch's shape (captured on first use; BatchGraph).  Same results."""
        assert self.n_done + batch.E <= self.max_episodes
        g = self.__dict__.get("_graph")
        if g is None or not g.matches(batch):
            g = self.__dict__["_graph"] = BatchGraph(self, batch, grad_sink, loss_weight)
            self._graph_sink = [t.data_ptr() for t in grad_sink]
        assert self._graph_sink == [t.data_ptr() for t in grad_sink], "the captured batch writes into the sink it was captured with"
        self.rec_index.fill_(2 * self.n_done)
        out = g.replay(batch)
        self.n_done += batch.E
        return out

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


class BatchGraph:
    """The training launch sequence of ONE batch shape (head_train.explicit_train_batch: ~450 launches for 32 episodes)
    frozen into a hipGraph and replayed once per step: the same kernels with the same arguments in the same order, so the
    same results bit for bit (tests/test_gpu_batched.py), without ~450 host launch calls per step and without the queue
    running dry behind the step's host wait (status words, Adam).  What a frozen sequence needs (as episode_graph.py):
      * inputs live in static buffers (the step's episodes are copied in: 29 MB at workload S, 32 episodes);
      * the attention-dropout seed advances in device memory; BatchNorm statistics are recorded (the runner's records);
      * the CG loops are captured with `lp_budget` iterations, of which only the first 1.5 x (slowly decaying maximum seen)
        + 8 stay enabled (r3d_graph_set_lp_budget: disabled kernel nodes are empty); a step that needs more reports "not
        converged" through the runner's counters like any other miss, and the trainer redoes it eagerly;
      * the folded q | k | v matrix is refreshed in place before a replay (the graph holds its address);
      * the packed-weight scratch of the point-wise GEMM may be used by the captured launches (r3d_set_wpack_in_capture):
        this graph is the only user of its stream's scratch while it replays."""

    def __init__(self, runner, example, grad_sink, loss_weight=0.1, lp_budget=None):
        import ctypes
        from . import _lib
        from .mpti import EpisodeSlot
        self.runner, self.model = runner, runner.model
        m = self.model
        self.E = example.E
        dev = runner.dev
        clone = lambda t: t.clone() if t is not None else None
        # static inputs: the same layouts as the example (point-major views stay point-major views)
        self.batch = EpisodeBatch(self._clone_x(example.support_x), example.support_y.clone(), self._clone_x(example.query_x),
                                  example.query_y.clone(), clone(example.gt_support_y), clone(example.gt_query_y),
                                  clone(example.support_flag))
        self.lp_budget = int(lp_budget if lp_budget is not None else min(m.lp_max_iter, 96))
        self.active_budget = self.lp_budget
        self._mx_decay = 0
        slot = EpisodeSlot(7000 + self.E)
        slot.fixed_budget = self.lp_budget
        slot.seed_dev = torch.full((1,), 104729, device=dev, dtype=torch.int32)
        self.slot = slot
        self.stream = torch.cuda.Stream()
        lib = _lib.load()
        saved_slot, saved_rec = m._slot, T.bn_recorder
        buffers = {k: v.clone() for k, v in m.named_buffers()}  # warm-up passes must not count in the running statistics
        sink_backup = [g.clone() for g in grad_sink]
        counters_backup = runner.counters.clone()
        old = lib.r3d_set_wpack_in_capture(1)
        try:
            m._slot = slot
            m._lp_force = False
            T.bn_recorder = runner.bn_records
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                for _ in range(2):  # eager warm-up on the capture stream: allocations, head buffers, the W scratch
                    self._once(grad_sink, loss_weight)
            cur.wait_stream(self.stream)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
                self._once(grad_sink, loss_weight)
            self.graph.instantiate()
        finally:
            lib.r3d_set_wpack_in_capture(old)
            m._slot, T.bn_recorder = saved_slot, saved_rec
        with torch.no_grad():
            for k, v in m.named_buffers():
                v.copy_(buffers[k])
            for g, b in zip(grad_sink, sink_backup):
                g.copy_(b)
            runner.counters.copy_(counters_backup)
        n_cg = ctypes.c_int(0)
        _lib.check(lib.r3d_graph_set_lp_budget(ctypes.c_void_p(self.graph.raw_cuda_graph()),
                                               ctypes.c_void_p(self.graph.raw_cuda_graph_exec()), self.lp_budget, ctypes.byref(n_cg)))
        assert n_cg.value > 0, "no CG nodes found in the captured batch"
        torch.cuda.synchronize()

    @staticmethod
    def _clone_x(x):
        if ops.is_point_major_view(x):
            return x.transpose(-1, -2).contiguous().transpose(-1, -2)
        return x.clone()

    def _once(self, grad_sink, loss_weight):
        self.out = explicit_train_batch(self.model, self.batch, grad_sink, loss_weight)
        self.runner._count(backward=True)

    def matches(self, batch):
        b = self.batch
        return (batch.E == b.E and batch.support_x.shape == b.support_x.shape and batch.query_x.shape == b.query_x.shape
                and (batch.support_flag is None) == (b.support_flag is None)
                and ops.is_point_major_view(batch.support_x) == ops.is_point_major_view(b.support_x))

    def set_lp_budget(self, budget):
        import ctypes
        from . import _lib
        budget = max(1, min(int(budget), self.lp_budget))
        if budget == self.active_budget:
            return
        torch.cuda.current_stream().synchronize()  # never edit an executable graph that is in flight
        _lib.check(_lib.load().r3d_graph_set_lp_budget(ctypes.c_void_p(self.graph.raw_cuda_graph()),
                                                       ctypes.c_void_p(self.graph.raw_cuda_graph_exec()), budget, None))
        self.active_budget = budget

    def adapt(self, status):
        """The CG budget of the next replays from a finished step's (bad, overflow, iterations, max)."""
        bad, _, _, mx = status
        if bad:
            self._mx_decay = max(self._mx_decay, mx)
            self.set_lp_budget(self.lp_budget)
        elif mx > 0:
            self._mx_decay = max(mx, self._mx_decay - max(1, self._mx_decay // 16))
            self.set_lp_budget(max(24, 8 * ((self._mx_decay + self._mx_decay // 2 + 8 + 7) // 8)))

    def replay(self, batch):
        """One training pass of `batch` (same shapes as the example) on the current stream; returns what
        explicit_train_batch returns (static tensors: valid until the next replay)."""
        m, b = self.model, self.batch
        with torch.no_grad():
            # (the launch sequence reads the clouds through x_all alone; masks, labels and flags through their own tensors)
            for dst, src in ((b.x_all, batch.x_all), (b.support_y, batch.support_y), (b.query_y, batch.query_y),
                             (b.gt_support_y, batch.gt_support_y), (b.gt_query_y, batch.gt_query_y),
                             (b.support_flag, batch.support_flag)):
                if dst is not None and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            if getattr(m, "use_attention", False):
                m.att_learner._fold()  # in place: the graph holds the folded matrix's address
        saved_slot = m._slot
        m._slot = self.slot  # (lp_converged() and the status counters read the slot's head buffers)
        try:
            self.graph.replay()
        finally:
            m._slot = saved_slot
        self.model.__dict__["_last_batch_slot"] = self.slot
        return self.out


def collate(episodes, batch_size):
    """Episode lists (train or test layout) -> EpisodeBatch objects of up to batch_size episodes."""
    return [EpisodeBatch.from_episodes(episodes[i:i + batch_size]) for i in range(0, len(episodes), batch_size)]
