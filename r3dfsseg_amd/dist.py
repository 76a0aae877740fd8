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
    needs a single all-reduce (SUM) followed by a division by the global episode count."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # one extra slot carries the episode count, so gradients AND their divisor travel in one all-reduce
        self.store = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.flat = self.store[:n]
        off = 0
        for p in self.params:  # p.grad becomes a view into the bucket
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, n_local_episodes):
        """Sum gradients over ranks and divide by the total number of episodes of the step."""
        total = self.store[-1:]
        total.fill_(float(n_local_episodes))  # a fill kernel: no host->device copy, no host sync
        if dist.is_initialized():  # also with one rank (cheap, and it keeps the single-rank path identical)
            dist.all_reduce(self.store, op=dist.ReduceOp.SUM)
        self.flat.div_(total)
        return self.flat


def all_reduce_histogram(hist):
    """Sum the (3, n_classes) int64 TP/GT/P histogram of the mIoU accumulator over ranks."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist
