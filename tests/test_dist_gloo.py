"""world_size-2 gloo test of the data-parallel helpers (episode sharding + single flat-bucket
gradient all-reduce), the N > 1 path of SURVEY.md 8e, on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from r3dfsseg_amd import dist as D
    assert D.init("gloo") == world
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 3)
    bucket = D.FlatGradBucket(lin.parameters())
    episodes = D.shard_episodes(5, rank, world)
    bucket.zero_()
    for e in episodes:  # gradient of a per-episode loss, accumulated into the flat bucket
        x = torch.full((2, 7), float(e + 1))
        lin(x).sum().backward()
    n_failed = bucket.all_reduce_mean(len(episodes))
    flat = bucket.flat.clone()
    hist = D.all_reduce_histogram(torch.tensor([[rank + 1, 2, 3]], dtype=torch.int64))
    # the failure flag travels with the gradients: one rank without an exact gradient is seen by every rank
    n_failed2 = bucket.all_reduce_mean(len(episodes), failed=(rank == 1))
    # BatchNorm running statistics: per rank during training, averaged over the ranks on request
    bn = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        bn[1].running_mean.fill_(float(rank))
        bn[1].running_var.fill_(1.0 + 2.0 * rank)
    n_sync = D.sync_running_stats(bn)
    out[rank] = (episodes, flat, hist, [p.grad.data_ptr() for p in lin.parameters()], bucket.flat.data_ptr(),
                 (n_failed, n_failed2), (n_sync, bn[1].running_mean.clone(), bn[1].running_var.clone()))
    dist.destroy_process_group()


def test_two_rank_flat_bucket_allreduce():
    world, port = 2, _free_port()
    with mp.get_context("spawn").Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    assert sorted(out[0][0] + out[1][0]) == [0, 1, 2, 3, 4] and not set(out[0][0]) & set(out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    # reference value: mean over the 5 episodes of the single-process gradient
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 3)
    for e in range(5):
        lin(torch.full((2, 7), float(e + 1))).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]) / 5
    assert torch.allclose(out[0][1], want, rtol=1e-6, atol=1e-6)
    assert out[0][2].tolist() == [[3, 4, 6]]
    assert out[0][3][0] == out[0][4]  # gradients are views into the one bucket
    assert out[0][5] == (0, 1) and out[1][5] == (0, 1)
    for r in (0, 1):
        n_sync, rm, rv = out[r][6]
        assert n_sync == 8 and torch.equal(rm, torch.full((4,), 0.5)) and torch.equal(rv, torch.full((4,), 2.0))
