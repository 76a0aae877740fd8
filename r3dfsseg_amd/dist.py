"""Data-parallel episode sharding.  One process per GPU; episodes are independent
units, so ranks take disjoint episodes and the ONLY collective is one all-reduce of a single flat
fp32 gradient bucket (376 896 floats = 1.5 MB, latency bound) per optimiser step, plus the tiny
TP/GT/P histogram reduce of the mIoU accumulator.  Backend "nccl" is RCCL on ROCm; "gloo" in CPU tests.
The reference has no distributed code (models/mpti_learner.py:24 is a commented-out DataParallel)."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    return world


def shard_episodes(n_episodes, rank, world):
    """Round-robin episode ids of this rank (every id exactly once across ranks)."""
    return list(range(rank, n_episodes, world))


class FlatGradBucket:
    """All trainable parameters' gradients viewed through ONE contiguous fp32 buffer, so a step
    needs a single all-reduce (SUM) followed by a division by the global episode count.  Two extra slots travel with the
    gradients: the episode count (the divisor) and a failure flag -- a rank whose episodes could not be solved exactly
    raises only AFTER the collective, together with every other rank (nobody is left waiting in an all-reduce)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.store = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self.flat = self.store[:n]
        off = 0
        for p in self.params:  # p.grad becomes a view into the bucket
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, n_local_episodes, failed=False):
        """Sum gradients over ranks and divide by the total number of episodes of the step.  `failed`: this rank has no
        exact gradient to contribute; returns the number of ranks that said so (a host read only when distributed)."""
        tail = self.store[-2:]
        tail[0].fill_(float(n_local_episodes))  # fill kernels: no host->device copy, no host sync
        tail[1].fill_(1.0 if failed else 0.0)
        n_failed = 1 if failed else 0
        if dist.is_initialized():  # also with one rank (cheap, and it keeps the single-rank path identical)
            dist.all_reduce(self.store, op=dist.ReduceOp.SUM)
            if dist.get_world_size() > 1:
                n_failed = int(tail[1].item())
        self.flat.div_(tail[0])
        return n_failed


def sync_running_stats(model):
    """Average the BatchNorm running statistics over the ranks (ONE all-reduce of ~2.6 k floats).  Ranks train on disjoint
    episodes, so their running statistics drift apart while their weights stay identical; call this before an evaluation
    sweep or a checkpoint so that every rank evaluates / saves the same model.  The reference is single-process
    (models/mpti_learner.py:24 is a commented-out DataParallel): there is no precedent to follow, the mean is the
    estimate that uses every rank's episodes."""
    bufs = [b for n, b in model.named_buffers() if n.endswith("running_mean") or n.endswith("running_var")]
    if not bufs or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for mod in model.modules():  # BatchNorm-folded weights are derived data
        if hasattr(mod, "_folded") and mod._folded is not None:
            mod._folded = (None, mod._folded[1])
    model.__dict__["_rank_local_stats"] = False
    return flat.numel()


def mark_rank_local_stats(model):
    """Called by the trainer after a multi-rank step: this rank's running statistics now hold episodes the other ranks'
    do not.  warn_rank_local_stats() -- used where a model is saved or evaluated -- then says so once, instead of N ranks
    silently saving / evaluating N different models (the fix is a collective, which only the caller can place where every
    rank reaches it: DPTrainer.sync_running_stats())."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        model.__dict__["_rank_local_stats"] = True


def warn_rank_local_stats(model, what):
    if model.__dict__.get("_rank_local_stats"):
        import warnings
        warnings.warn("%s with per-rank BatchNorm running statistics: call DPTrainer.sync_running_stats() (on every rank) "
                      "first, or the ranks %s different models" % (what, "save" if "checkpoint" in what else "evaluate"))
        model.__dict__["_rank_local_stats"] = False  # (said once per training phase)


def all_reduce_histogram(hist):
    """Sum the (3, n_classes) int64 TP/GT/P histogram of the mIoU accumulator over ranks."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist
