"""Data-parallel training step: episodes are sharded over ranks, every rank runs
forward + backward of its own episodes into ONE flat fp32 gradient bucket, a single RCCL all-reduce
(SUM) over xGMI follows and every rank applies the same Adam step.  The reference itself steps Adam
after every single episode (mpti_train_noise.py:57,98); with world_size 1 and one episode per step
this class does exactly that.  BatchNorm running statistics are per rank during training (ranks see disjoint
episodes) and are averaged over the ranks by DPTrainer.sync_running_stats() before evaluation / checkpoints."""
import torch

from . import dist as D


class SolverMiss(RuntimeError):
    """An episode whose label propagation (forward or adjoint CG) did not converge even on the conservative schedule: the
    one failure a rank reports through the gradient bucket's failure slot instead of raising before the collective."""


class DPTrainer:
    def __init__(self, learner, n_slots=0, example=None, lp_budget=None, batch_size=0, guard_every=64, batch_graph=False):
        """batch_size > 0 (the throughput path, batched.py): the local episodes of a step go through ONE launch sequence
        per batch of up to batch_size episodes -- every kernel works on the whole batch, the weight gradients are summed
        inside the dW GEMMs straight into the bucket.
        batch_graph (with batch_size > 0): a step whose local episodes are ONE batch of that size replays the batch's
        launch sequence as a captured hipGraph (batched.BatchGraph; captured at the first such step): same kernels, same
        results, no per-launch host work.
        n_slots > 0 (round 2's schedule, kept for the single-episode learner): the episodes are replayed as captured
        hipGraphs, n_slots in flight on separate HIP streams (episode_graph.EpisodeGraphs; `example` = one episode fixing
        the shapes); every slot accumulates into its own gradient row and the rows are summed into the bucket.
        Neither: eager launches, one episode after the other (the reference's schedule)."""
        # guard_every (n_slots > 1 only): every that many steps the slots' label-propagation edge weights are recomputed
        # on the drained chip and compared bit for bit (EpisodeGraphs.verify_graph_weights): kernels of several streams
        # share the chip in that schedule, and wrong weights next to bf16-MFMA-dense waves are what round 2 found
        # before the library was rebuilt without packed fp32 arithmetic.  0 switches the guard off.
        self.guard_every = guard_every
        self.n_steps = 0
        self.learner = learner
        self.model = learner.model
        self.bucket = D.FlatGradBucket(self.model.parameters())
        self.graphs = None
        self.runner = None
        self.batch_size = batch_size
        self.batch_graph = bool(batch_graph) and batch_size > 0
        self.last_outputs = []  # per-episode results of the last step: (loss, lp_loss, contrast_loss, logits (n_q, C, N), metrics (4,))
        self.redone = False   # did the last step fall back to the conservative schedule?
        self.n_redone = 0     # ... and how many steps did so far
        self.last_status = (0, 0, 0, 0)
        if batch_size:
            from .batched import EpisodeBatchRunner
            self.runner = EpisodeBatchRunner(self.model)
        if n_slots:
            from .episode_graph import EpisodeGraphs
            self.rows = torch.zeros(n_slots, self.bucket.store.numel(), device=self.bucket.store.device)
            self.graphs = EpisodeGraphs(self.model, example, n_slots, train=True, lp_budget=lp_budget, grad_rows=self.rows)

    def step(self, episodes, logger=None):
        """episodes: list of train-layout data lists (loader.py:1666-1671) local to this rank.
        Returns the mean (lp_loss + 0.1 contrast) over the local episodes as a device tensor.

        Fails closed: the optimiser never steps on a gradient from an episode whose CG solves (forward or adjoint) ran
        out of launch budget, whose 201-NN survivor buffer overflowed or whose one-launch FPS timed out.  The status
        words are read after the local episodes and BEFORE the all-reduce (one host wait per step, ~0.3 % of a
        32-episode step); a miss makes this rank redo its episodes of the step on the conservative schedule (full
        budget, exact kernels).  Ranks decide locally: a rank's contribution to the all-reduce is an exact gradient or
        nothing -- if the conservative schedule fails too, the rank says so in the bucket's failure slot, takes part in
        the collective like everybody else, and EVERY rank raises after it (no rank is left blocked in an all-reduce)."""
        self.model.train()
        self.redone = False
        if self.runner is not None:
            from .batch import EpisodeBatch
            from .batched import collate
            batches = episodes if (episodes and isinstance(episodes[0], EpisodeBatch)) else collate(episodes, self.batch_size)
            n_local = sum(b.E for b in batches)
            self.bucket.zero_()
            self.runner.begin_step()
            total = None
            outs = []
            use_graph = self.batch_graph and len(batches) == 1 and batches[0].E == self.batch_size
            if use_graph and self.runner.__dict__.get("_graph") is None:
                try:  # capture on first use; a stack that cannot capture this sequence keeps launching it eagerly
                    from .batched import BatchGraph
                    sink = [p.grad for p in self.bucket.params]
                    self.runner._graph = BatchGraph(self.runner, batches[0], sink)
                    self.runner._graph_sink = [t.data_ptr() for t in sink]
                    self.bucket.zero_()
                    self.runner.begin_step()
                except Exception as exc:  # noqa: BLE001 -- whatever the capture ran into, the eager path is the same computation
                    import warnings
                    warnings.warn("batch graph capture failed (%r): the batched step stays on eager launches" % (exc,))
                    self.batch_graph = use_graph = False
                    torch.cuda.synchronize()
                    self.bucket.zero_()
                    self.runner.begin_step()
            for b in batches:
                o = (self.runner.train_batch_graph if use_graph else self.runner.train_batch)(b, [p.grad for p in self.bucket.params])
                outs += [(o[0][e], o[3][e], o[4][e], o[1][e], o[2][e]) for e in range(b.E)]
                loss = o[0].sum()
                total = loss if total is None else total + loss
            self.last_status = self.runner.step_status()
            if use_graph:
                self.runner._graph.adapt(self.last_status)
            failed, apply_stats = None, self.runner.apply_running_stats
            if self.last_status[0] or self.last_status[1]:
                eps = [b.episode(e) for b in batches for e in range(b.E)]
                try:
                    total, outs, apply_stats = self._eager_pass(eps, logger, conservative=True)
                except SolverMiss as exc:
                    failed, apply_stats = exc, None
                self.redone = True
                self.n_redone += 1
            self.last_outputs = outs
            # (the running statistics only move once the collective has said that no rank failed: an abandoned step leaves
            # neither Adam nor the BatchNorm buffers touched)
            self._reduce_and_step(n_local, failed, apply_stats)
            return total / max(n_local, 1)
        failed, apply_stats = None, None
        self.last_outputs = []
        try:
            if self.graphs is not None:
                total = self.graphs.run(episodes, apply_bn=False)
                bad, overflow, _, _ = self.graphs.step_status()
                self.n_steps += 1
                if self.guard_every and self.graphs.n_slots > 1 and self.n_steps % self.guard_every == 0:
                    wrong = self.graphs.verify_graph_weights()
                    if wrong:
                        raise RuntimeError("guard: %d label-propagation edge weights computed beside other streams' kernels "
                                           "differ from their recomputation on the idle chip" % wrong)
                if bad or overflow:
                    self.redone = True
                    self.n_redone += 1
                    total, self.last_outputs, apply_stats = self._eager_pass(episodes, logger, conservative=True)
                else:
                    n_ep = len(episodes)
                    apply_stats = lambda: self.graphs.apply_running_stats(n_ep)
                    torch.sum(self.rows, 0, out=self.bucket.store)
            else:
                total, self.last_outputs, apply_stats = self._eager_pass(episodes, logger, conservative=False)
        except SolverMiss as exc:  # (anything else -- a launch failure, a guard mismatch -- is not recoverable: it propagates)
            failed, total, apply_stats = exc, torch.zeros((), device=self.bucket.store.device), None
        self._reduce_and_step(len(episodes), failed, apply_stats)
        return total / max(len(episodes), 1)

    def _reduce_and_step(self, n_local, failed, apply_stats=None):
        """The step's ONE collective (gradients + episode count + failure flag), then the BatchNorm running statistics of
        the step's episodes and Adam -- or, if any rank could not produce an exact gradient, the same error on every rank
        with neither touched."""
        n_failed = self.bucket.all_reduce_mean(n_local, failed=failed is not None)
        if n_failed:
            raise RuntimeError("training step abandoned on all ranks: %d rank(s) could not solve their episodes exactly%s" % (
                n_failed, (" (this rank: %s)" % failed) if failed is not None else "")) from failed
        if apply_stats is not None:
            apply_stats()
            D.mark_rank_local_stats(self.model)
        self.learner.optimizer.step()
        self.learner.lr_scheduler.step()

    def sync_running_stats(self):
        """Average the BatchNorm running statistics over the ranks (dist.sync_running_stats): call before an evaluation
        sweep or a checkpoint, so that every rank evaluates / saves the same model."""
        return D.sync_running_stats(self.model)

    def _eager_pass(self, episodes, logger, conservative):
        """Forward + backward of every episode into the bucket (parameter .grad tensors are views into it).  An attempt
        on the adaptive schedule that misses (CG budget, 201-NN overflow, FPS time-out) is discarded -- its gradient AND
        its BatchNorm statistics -- and the episode is redone on the conservative schedule.  Returns (sum of the losses,
        per-episode results, a callable that folds the kept attempts' BatchNorm statistics into the running buffers, episode
        after episode: the caller runs it once the step is known to be kept)."""
        from . import train_ops as T
        self.bucket.zero_()
        total = None
        outs = []
        rec = T.BNRecorder(len(episodes), self.bucket.store.device)
        rec.index_dev = torch.zeros(1, device=self.bucket.store.device, dtype=torch.int32)
        for n_kept, data in enumerate(episodes):
            (support_x, support_y, query_x, query_y, support_c, query_c, gt_support_y, gt_query_y, bg_pcd_x, bg_pcd_y,
             support_flag) = data
            for lp_iters in ((self.model.lp_max_iter,) if conservative else (None, self.model.lp_max_iter)):
                keep = self.bucket.flat.clone() if lp_iters is None and total is not None else None
                rec.index_dev.fill_(2 * n_kept)  # (a discarded attempt's records are overwritten by the attempt that is kept)
                saved, T.bn_recorder = T.bn_recorder, rec
                try:
                    out = self.model(support_x, support_y, query_x, query_y, gt_support_y=gt_support_y,
                                     gt_query_y=gt_query_y, train=True, logger=logger, support_flag=support_flag,
                                     lp_iters=lp_iters)
                    loss = out[1] + 0.1 * out[2]  # mpti_learner.py:66
                    loss.backward()               # accumulates into the bucket views
                finally:
                    T.bn_recorder = saved
                if self.model.lp_converged(backward=True):
                    break
                if lp_iters is None:              # drop the inexact gradient again, then the conservative schedule
                    if keep is not None:
                        self.bucket.flat.copy_(keep)
                    else:
                        self.bucket.zero_()       # (first episode of the step: nothing to keep)
            else:
                raise SolverMiss("label propagation did not converge in %d CG iterations" % self.model.lp_max_iter)
            total = loss.detach() if total is None else total + loss.detach()
            outs.append((loss.detach(), out[1].detach(), out[2].detach(), out[0].detach(), torch.stack([torch.as_tensor(v, device=loss.device, dtype=torch.float32) for v in out[3:]])))
        n = len(episodes)
        return total, outs, (lambda: rec.apply(n))
