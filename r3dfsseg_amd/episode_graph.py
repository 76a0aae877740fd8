"""Episodes in flight: one captured hipGraph per slot, G slots on G HIP streams.

One episode of this workload is ~900 small-to-medium kernel launches whose grids mostly under-fill 256 CUs
(CG iterations, FPS rounds, BatchNorm reductions), so a single in-order stream leaves the chip idle between
and beside them.  Episodes are independent units, so the MI355X-native schedule is:

  * freeze the launch sequence of ONE episode (forward, or forward + backward written out without the
    autograd engine: head_train.explicit_train_episode) into a hipGraph -- the host
    then spends ~0.3 ms per episode instead of ~9 ms of Python/ctypes launch calls;
  * keep G such graphs, each with private activations, head buffers (mpti.EpisodeSlot) and -- when training --
    a private row of the flat gradient bucket, and replay them on G streams so the GPU overlaps them;
  * join the streams once per step; the trainer then sums the G gradient rows into the rank's bucket and
    issues the single RCCL all-reduce (dp_train.DPTrainer).

The reference runs one episode at a time in eager PyTorch (mpti_train_noise.py:57-98, eval_noise.py:46-70);
per-episode results are the same as the eager path of this package (tests/test_gpu_graph.py).

Frozen-sequence consequences, all checked or documented:
  * the CG launch budget is fixed at capture (iterations after convergence are no-op launches); every replay
    adds its "did not converge / survivor buffer overflowed" flags into ``slot.bad`` and ``check()`` reads them;
  * the attention-dropout seed advances in device memory (slot.seed_dev), not in a Python counter;
  * BatchNorm running statistics are updated by slot 0 only (shared buffers, concurrent slots would race);
    the batch statistics used for normalisation are per episode either way.
"""
import os

import torch

from . import ops, train_ops
from .mpti import EpisodeSlot


class _Slot:
    pass


class EpisodeGraphs:
    def __init__(self, model, example, n_slots=4, train=False, lp_budget=None, grad_rows=None, loss_weight=0.1):
        """example: one episode (list of tensors, train layout loader.py:1666-1671 or the 4-tensor test layout
        (support_x, support_y, query_x, query_y)) fixing shapes and dtypes.  grad_rows: (n_slots, n_params[+1])
        fp32 buffer, row s receives slot s's gradients (train only)."""
        self.model, self.train, self.n_slots = model, train, n_slots
        self.loss_weight = loss_weight
        dev = next(model.parameters()).device
        self.params = [p for p in model.parameters() if p.requires_grad]
        if train:
            n = sum(p.numel() for p in self.params)
            self.grad_rows = grad_rows if grad_rows is not None else torch.zeros(n_slots, n, device=dev)
            assert self.grad_rows.shape[0] == n_slots and self.grad_rows.shape[1] >= n
        if lp_budget is None:  # CG iterations frozen into the graph; launches after convergence are no-ops (~3 us each)
            lp_budget = min(model.lp_max_iter, 200 if train else 128)
        self.lp_budget = lp_budget
        self.slots = []
        self.ev_start = torch.cuda.Event()
        saved_slot = model._slot
        buffers = {k: v.clone() for k, v in model.named_buffers()}  # warm-up passes must not leak into BN statistics
        was_training = model.training
        model.train(train)
        try:
            for s in range(n_slots):
                self.slots.append(self._capture(s, example, dev))
        finally:
            model._slot = saved_slot
            train_ops.update_running_stats = True
            model.train(was_training)
        with torch.no_grad():
            for k, v in model.named_buffers():
                v.copy_(buffers[k])
        self.reset()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ capture
    def _run_once(self, sl):
        model = self.model
        if self.train:
            from .head_train import explicit_train_episode
            loss, sl.logits, _ = explicit_train_episode(model, sl.inputs, sl.grad_views, self.loss_weight)
            sl.loss_sum += loss
            hb = model._slot.last[1]
            ok = hb.stats[0] * hb.stats_bwd[0].clamp(max=1)
        else:
            sx, sy, qx, qy = sl.inputs[:4]
            with torch.no_grad():
                logits, loss = model(sx, sy, qx, qy)
            sl.loss_sum += loss
            hb = model._slot.last[1]
            ok = hb.stats[0]
            sl.logits = logits
        sl.bad += 1 - ok + hb.desc[ops.HD_FPS_TIMEOUT].clamp(max=1)
        sl.knn_overflow += hb.knn_status[0].clamp(max=1)
        sl.cg_iters += hb.stats[1]
        torch.maximum(sl.cg_max, hb.stats[1], out=sl.cg_max)

    def _capture(self, s, example, dev):
        model = self.model
        sl = _Slot()
        sl.stream = torch.cuda.Stream()
        sl.done = torch.cuda.Event()
        sl.inputs = [t.to(dev).clone() for t in example]
        sl.loss_sum = torch.zeros((), device=dev)
        sl.bad = torch.zeros((), device=dev, dtype=torch.int32)
        sl.knn_overflow = torch.zeros((), device=dev, dtype=torch.int32)
        sl.cg_iters = torch.zeros((), device=dev, dtype=torch.int32)
        sl.cg_max = torch.zeros((), device=dev, dtype=torch.int32)
        st = EpisodeSlot(s)
        st.fixed_budget = self.lp_budget
        st.update_running = (s == 0)
        # one-launch FPS only while all slots' FPS grids fit the chip together (head_proto.hip, 2b)
        fps_blocks = (model.n_way * model.k_shot * model.n_points + 255) // 256 + model.n_way + 1
        # 2 workgroups of the one-launch FPS fit a CU at D <= 192 (234 VGPRs), 1 above: 512 / 256 slots on the chip.
        # HIP runs the streams on GPU_MAX_HW_QUEUES (default 4) hardware queues, one kernel at a time per queue, so at
        # most that many FPS grids are ever resident together, whatever the number of slots
        in_flight = min(self.n_slots, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
        st.fps_one_launch = in_flight * fps_blocks <= (500 if model.feat_dim <= 192 else 250)
        if self.train:
            st.seed_dev = torch.full((1,), 7919 * (s + 1), device=dev, dtype=torch.int32)
            off, sl.grad_views = 0, []
            for p in self.params:  # slot s adds its parameter gradients into row s
                sl.grad_views.append(self.grad_rows[s, off:off + p.numel()].view_as(p))
                off += p.numel()
        sl.state = st
        model._slot = st
        train_ops.update_running_stats = st.update_running
        cur = torch.cuda.current_stream()
        sl.stream.wait_stream(cur)
        with torch.cuda.stream(sl.stream):
            for _ in range(2):  # eager warm-up: allocations, head buffers, lazily initialised library state
                self._run_once(sl)
        cur.wait_stream(sl.stream)
        torch.cuda.synchronize()
        sl.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: only this thread launches into the capture (no autograd engine threads in the
        # explicit episode), while other threads -- e.g. the RCCL watchdog of torch.distributed polling its events --
        # must stay free to call the HIP runtime
        with torch.cuda.graph(sl.graph, capture_error_mode="thread_local"):
            self._run_once(sl)
        return sl

    # ------------------------------------------------------------------ replay
    def reset(self):
        """Zero the per-step accumulators (gradient rows, loss sums).  Convergence counters are kept."""
        if self.train:
            self.grad_rows.zero_()
        for sl in self.slots:
            sl.loss_sum.zero_()

    def _refresh_folds(self):
        """The graphs hold pointers to the folded weights (eval: BatchNorm folded into the GEMM epilogues; train:
        the fused q|k|v matrix); after a weight update they are recomputed IN PLACE (dgcnn._refresh), once per
        step, on the stream the slots wait for."""
        m = self.model
        if getattr(m, "use_attention", False):
            m.att_learner._fold()
        if not self.train:
            m.encoder._fold()
            m.base_learner._fold()

    def run(self, episodes, logits_out=None):
        """Replay one graph per episode, round-robin over the slots, and join the slot streams into the current
        stream.  logits_out: optional (len(episodes), n_q, n_classes, N) buffer receiving every episode's query
        logits.  Returns the device scalar sum of the episodes' losses."""
        main = torch.cuda.current_stream()
        self.reset()
        self._refresh_folds()
        self.ev_start.record(main)
        G = self.n_slots
        for e, ep in enumerate(episodes):
            sl = self.slots[e % G]
            with torch.cuda.stream(sl.stream):
                if e < G:
                    sl.stream.wait_event(self.ev_start)
                for dst, src in zip(sl.inputs, ep):
                    dst.copy_(src, non_blocking=True)
                    if src.is_cuda:
                        src.record_stream(sl.stream)
                sl.graph.replay()
                if logits_out is not None:
                    logits_out[e].copy_(sl.logits, non_blocking=True)
        for sl in self.slots[:min(G, len(episodes))]:
            sl.done.record(sl.stream)
            main.wait_event(sl.done)
        total = self.slots[0].loss_sum
        for sl in self.slots[1:min(G, len(episodes))]:
            total = total + sl.loss_sum
        return total

    def check(self):
        """Host check (synchronises): (number of replays whose label propagation did not converge or whose
        201-NN survivor buffer overflowed since the last check, mean CG iterations, max CG iterations)."""
        self.last_unconverged = int(sum(int(sl.bad.item()) for sl in self.slots))
        self.last_knn_overflow = int(sum(int(sl.knn_overflow.item()) for sl in self.slots))
        bad = self.last_unconverged + self.last_knn_overflow
        it = [int(sl.cg_iters.item()) for sl in self.slots]
        mx = max(int(sl.cg_max.item()) for sl in self.slots)
        for sl in self.slots:
            sl.bad.zero_(); sl.knn_overflow.zero_(); sl.cg_iters.zero_(); sl.cg_max.zero_()
        return bad, sum(it), mx
