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
    adds its "did not converge / survivor buffer overflowed" flags into per-step device counters: ``step_status()``
    (one host wait, used by the trainer before every optimiser step) and ``check()`` read them;
  * the attention-dropout seed advances in device memory (slot.seed_dev), not in a Python counter;
  * BatchNorm running statistics: a replay RECORDS its batch statistics (train_ops.BNRecorder: shared buffers, concurrent
    slots would race) and the owner applies the records in episode order after the step -- bit for bit the running
    statistics of the reference's one-episode-at-a-time schedule (tests/test_gpu_graph.py).

The CG launch budget is adaptive without re-capturing: launches after convergence return at once but still cost
~2.5 us of queue time each, so ``run()`` reads -- one step late, through a pinned buffer, never blocking -- the largest
iteration count seen and keeps only the CG kernel nodes of the first ``1.5 * max + 8`` iterations enabled
(``r3d_graph_set_lp_budget``: a disabled node is an empty node).  A replay that needs more reports "not converged"
through ``check()`` as before, and the budget returns to the captured maximum.
"""
import ctypes
import os

import torch

from . import _lib, ops, train_ops
from .mpti import EpisodeSlot


class _Slot:
    pass


class EpisodeGraphs:
    def __init__(self, model, example, n_slots=4, train=False, lp_budget=None, grad_rows=None, loss_weight=0.1,
                 eval_flag=False):
        """example: one episode (list of tensors, train layout loader.py:1666-1671 or the 4-tensor test layout
        (support_x, support_y, query_x, query_y)) fixing shapes and dtypes.  grad_rows: (n_slots, n_params[+1])
        fp32 buffer, row s receives slot s's gradients (train only)."""
        self.model, self.train, self.n_slots = model, train, n_slots
        self.loss_weight = loss_weight
        self.eval_flag = eval_flag          # eval graphs: forward(..., eval=True), the clean-shot detection of eval_noise.py
        dev = next(model.parameters()).device
        self.params = [p for p in model.parameters() if p.requires_grad]
        if train:
            n = sum(p.numel() for p in self.params)
            self.grad_rows = grad_rows if grad_rows is not None else torch.zeros(n_slots, n, device=dev)
            assert self.grad_rows.shape[0] == n_slots and self.grad_rows.shape[1] >= n
        if lp_budget is None:  # CG iterations frozen into the graph; the ones beyond the adaptive budget are disabled nodes
            lp_budget = min(model.lp_max_iter, 200)
        self.lp_budget = lp_budget          # CG iterations captured into every graph
        self.active_budget = lp_budget      # ... of which this many are enabled
        self.adaptive_budget = os.environ.get("R3D_FIXED_LP_BUDGET") is None
        # per slot, per run(): [not converged / FPS time-out, 201-NN overflow, CG iterations (sum), CG iterations (max)]
        self.counters = torch.zeros(n_slots, 4, device=dev, dtype=torch.int32)
        self._probes = []                   # (event, pinned copy of the counters) of the runs not yet accounted for
        self._probe_pool = [torch.zeros(n_slots, 4, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._since_check = [0, 0, 0, 0]    # host totals since the last check(): bad, overflow, iterations, max
        self._mx_decay = 0                  # slowly decaying maximum of the iteration counts (budget adaptation)
        self.slots = []
        self.ev_start = torch.cuda.Event()
        self.max_episodes = 256             # episodes of one run() the BatchNorm records are sized for
        self.bn_records = train_ops.BNRecorder(self.max_episodes, dev) if train else None
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
            train_ops.bn_recorder = None
            model.train(was_training)
        with torch.no_grad():
            for k, v in model.named_buffers():
                v.copy_(buffers[k])
        self.reset()
        self.counters.zero_()  # the warm-up and capture passes are not episodes of any step
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ capture
    def _run_once(self, sl):
        model = self.model
        if self.train:
            from .head_train import explicit_train_episode
            loss, sl.logits, metrics, lp, con = explicit_train_episode(model, sl.inputs, sl.grad_views, self.loss_weight)
            sl.loss_sum += loss
            # the last episode's parts of the reference's train() tuple (mpti_learner.py:78), for the one-slot learner path
            sl.parts[0].copy_(lp); sl.parts[1].copy_(con)
            for i in range(4):
                sl.parts[2 + i].copy_(metrics[i])
            hb = model._slot.last[1]
            ok = hb.stats[0] * hb.stats_bwd[0].clamp(max=1)
        else:
            sx, sy, qx, qy = sl.inputs[:4]
            with torch.no_grad():
                logits, loss = model(sx, sy, qx, qy, eval=self.eval_flag)
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
        sl.parts = torch.zeros(6, device=dev)  # lp_loss, contrast loss, the four debug metrics of the last replay
        sl.bad, sl.knn_overflow, sl.cg_iters, sl.cg_max = (self.counters[s, i] for i in range(4))
        st = EpisodeSlot(s)
        st.fixed_budget = self.lp_budget
        sl.ep2 = torch.zeros(1, device=dev, dtype=torch.int32)  # 2 x (index of the episode inside its run())
        # one-launch FPS only while all slots' FPS grids fit the chip together (head_proto.hip, 2b)
        fps_blocks = (model.n_way * model.k_shot * model.n_points + 255) // 256 + model.n_way + 1
        # 2 workgroups of the one-launch FPS fit a CU at D <= 192 (234 VGPRs), 1 above: 512 / 256 slots on the chip.
        # HIP runs the streams on GPU_MAX_HW_QUEUES (default 4) hardware queues, one kernel at a time per queue, so at
        # most that many FPS grids are ever resident together, whatever the number of slots
        in_flight = min(self.n_slots, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
        st.fps_one_launch = (in_flight * fps_blocks <= (500 if model.feat_dim <= 192 else 250)
                             and os.environ.get("R3D_FPS_PER_ROUND") is None)  # A/B switch: one launch per FPS round
        if self.train:
            st.seed_dev = torch.full((1,), 7919 * (s + 1), device=dev, dtype=torch.int32)
            off, sl.grad_views = 0, []
            for p in self.params:  # slot s adds its parameter gradients into row s
                sl.grad_views.append(self.grad_rows[s, off:off + p.numel()].view_as(p))
                off += p.numel()
        sl.state = st
        model._slot = st
        if self.train:
            self.bn_records.index_dev = sl.ep2
            train_ops.bn_recorder = self.bn_records
        cur = torch.cuda.current_stream()
        sl.stream.wait_stream(cur)
        with torch.cuda.stream(sl.stream):
            for _ in range(2):  # eager warm-up: allocations, head buffers, lazily initialised library state
                self._run_once(sl)
        cur.wait_stream(sl.stream)
        torch.cuda.synchronize()
        sl.graph = torch.cuda.CUDAGraph(keep_graph=True)  # the hipGraph_t stays: its nodes are enabled / disabled later
        # thread-local capture mode: only this thread launches into the capture (no autograd engine threads in the
        # explicit episode), while other threads -- e.g. the RCCL watchdog of torch.distributed polling its events --
        # must stay free to call the HIP runtime
        with torch.cuda.graph(sl.graph, capture_error_mode="thread_local"):
            self._run_once(sl)
        sl.graph.instantiate()
        return sl

    # ------------------------------------------------------------------ CG launch budget
    def set_lp_budget(self, budget):
        """Enable the CG kernel nodes of the first `budget` iterations in every slot's graph, disable the rest."""
        budget = max(1, min(int(budget), self.lp_budget))
        if budget == self.active_budget:
            return
        lib = _lib.load()
        for sl in self.slots:
            sl.stream.synchronize()  # never edit an executable graph that is in flight
            n_cg = ctypes.c_int(0)
            _lib.check(lib.r3d_graph_set_lp_budget(ctypes.c_void_p(sl.graph.raw_cuda_graph()),
                                                   ctypes.c_void_p(sl.graph.raw_cuda_graph_exec()), budget, ctypes.byref(n_cg)))
            assert n_cg.value > 0, "no CG nodes found in the captured episode"
        self.active_budget = budget

    @staticmethod
    def budget_for(mx):
        """Enabled CG iterations for an observed maximum of `mx`: half as many again plus 8, rounded up to 8, at least 24
        (convergence is detected inside the last productive launch, so nothing extra is needed for the test itself)."""
        return max(24, 8 * ((mx + mx // 2 + 8 + 7) // 8))

    def _account(self, c):
        """One finished run()'s counters (host copy): totals for check(), launch budget for the next runs."""
        bad, ovf, its, mx = int(c[:, 0].sum()), int(c[:, 1].sum()), int(c[:, 2].sum()), int(c[:, 3].max())
        t = self._since_check
        t[0] += bad; t[1] += ovf; t[2] += its; t[3] = max(t[3], mx)
        self._probe_pool.append(c)
        if bad > 0:          # a replay did not converge: back to everything that was captured
            self._mx_decay = max(self._mx_decay, mx)
            self.set_lp_budget(self.lp_budget)
        elif self.adaptive_budget and mx > 0:  # the budget follows a slowly decaying maximum, so it can shrink again
            self._mx_decay = max(mx, self._mx_decay - max(1, self._mx_decay // 16))
            self.set_lp_budget(self.budget_for(self._mx_decay))
        return bad, ovf, its, mx

    def _adapt_budget(self):
        """At most two runs stay ahead of the GPU: wait for the run before the previous one and read the counters it
        left in pinned memory (no device synchronisation, the previous run keeps the GPU busy meanwhile)."""
        while len(self._probes) >= 2:
            ev, c = self._probes.pop(0)
            ev.synchronize()
            self._account(c)

    def step_status(self):
        """Host wait for the LATEST run(): (replays that did not converge or timed out, 201-NN overflows, sum and max
        of the CG iteration counts) of that run alone.  The trainer calls this before it steps the optimiser."""
        out = (0, 0, 0, 0)
        while self._probes:
            ev, c = self._probes.pop(0)
            ev.synchronize()
            out = self._account(c)
        return out

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

    def apply_running_stats(self, n_episodes):
        """Fold the BatchNorm batch statistics the last run() recorded into the running statistics, in episode order."""
        if self.train:
            self.bn_records.apply(n_episodes)

    def run(self, episodes, logits_out=None, apply_bn=True):
        """Replay one graph per episode, round-robin over the slots, and join the slot streams into the current
        stream.  logits_out: optional (len(episodes), n_q, n_classes, N) buffer receiving every episode's query
        logits.  apply_bn=False: the caller applies the recorded BatchNorm statistics itself (apply_running_stats), e.g.
        only once it knows that the step is kept.  Returns the device scalar sum of the episodes' losses."""
        assert len(episodes) <= self.max_episodes
        main = torch.cuda.current_stream()
        self._adapt_budget()
        self.reset()
        self.counters.zero_()  # on the main stream, which every slot stream waits for: counters are per run
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
                if self.train:
                    sl.ep2.fill_(2 * e)
                sl.graph.replay()
                if logits_out is not None:
                    logits_out[e].copy_(sl.logits, non_blocking=True)
        for sl in self.slots[:min(G, len(episodes))]:
            sl.done.record(sl.stream)
            main.wait_event(sl.done)
        if apply_bn:
            self.apply_running_stats(len(episodes))
        total = self.slots[0].loss_sum
        for sl in self.slots[1:min(G, len(episodes))]:
            total = total + sl.loss_sum
        c = self._probe_pool.pop()
        c.copy_(self.counters, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        self._probes.append((ev, c))
        return total

    def verify_graph_weights(self):
        """Runtime guard of the multi-stream schedule: with every stream drained, recompute the
        label-propagation edge weights of each slot's last episode alone on the chip and compare their bits with what the
        slot's solve used while other slots' kernels ran beside it.  Returns the number of differing entries (host int)."""
        from . import ops
        torch.cuda.synchronize()
        bad = 0
        for sl in self.slots:
            last = sl.state.last
            if last is not None:
                bad += int(ops.graph_weights_verify(last[1], self.model.sigma).item())
        return bad

    def check(self):
        """Host check (synchronises): (number of replays whose label propagation did not converge or whose
        201-NN survivor buffer overflowed since the last check, sum of the CG iterations, max CG iterations)."""
        self.step_status()
        bad, ovf, its, mx = self._since_check
        self._since_check = [0, 0, 0, 0]
        self.last_unconverged, self.last_knn_overflow = bad, ovf
        return bad + ovf, its, mx
