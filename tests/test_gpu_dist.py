"""The N > 1 training path with the REAL trainer: two ranks, each a process with its own model copy on
the one GPU of the test box (gloo between them), run DPTrainer.step on disjoint halves of a step's episodes; the
bucket after the all-reduce must be the single-process mean gradient over all the episodes, the weights after Adam
must agree between the ranks, and the mIoU histogram must reduce to the single-process one."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_EPISODES = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(slots, batch=0):
    from r3dfsseg_amd import synthetic as S
    from r3dfsseg_amd.dp_train import DPTrainer
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    model = MPTI_SelfAtten(SimpleNamespace(**cfg))
    model.load_state_dict(S.make_state_dict(cfg, 123))
    model.cuda().train()
    model.att_learner.dropout.p = 0.0  # the dropout seed is per slot / per call: parity needs it off
    eps = []
    for e in range(N_EPISODES):
        data, _ = S.make_episode(cfg, seed=60 + e, noise_ratio=0.5, train=True)
        eps.append([t.cuda() for t in data])
    learner = SimpleNamespace(model=model)
    learner.optimizer = torch.optim.Adam(
        [{'params': model.encoder.parameters(), 'lr': 0.0001}, {'params': model.base_learner.parameters()},
         {'params': model.att_learner.parameters()}, {'params': model.proj.parameters()}], lr=1e-3)
    learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
    trainer = DPTrainer(learner, n_slots=slots, example=eps[0], batch_size=batch)
    return cfg, model, eps, trainer


def _hist(model, eps, ids):
    from r3dfsseg_amd.metrics import MIoUAccumulator
    acc = MIoUAccumulator([3, 6, 9])
    model.eval()
    for e in ids:
        with torch.no_grad():
            logits, _ = model(*eps[e][:4], lp_iters=model.lp_max_iter)
        acc.update(logits.argmax(1), eps[e][3], [3, 6] if e % 2 == 0 else [6, 9])
    model.train()
    return acc


def _worker(rank, world, port, slots, out, batch=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from r3dfsseg_amd import dist as D
    assert D.init("gloo") == world
    torch.cuda.set_device(0)
    cfg, model, eps, trainer = _setup(slots, batch)
    ids = D.shard_episodes(N_EPISODES, rank, world)
    acc = _hist(model, eps, ids)      # before the step: both ranks hold the same weights
    acc.reduce()
    loss = trainer.step([eps[e] for e in ids])
    torch.cuda.synchronize()
    # ranks saw disjoint episodes: their BatchNorm running statistics differ until they are averaged
    own = model.encoder.conv.layer[1].running_mean.clone()
    n_synced = trainer.sync_running_stats()
    out[rank] = dict(ids=ids, grad=trainer.bucket.flat.cpu(), loss=float(loss), redone=trainer.redone,
                     params=torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu(),
                     hist=acc.hist.cpu(), own_stat=own.cpu(), n_synced=n_synced,
                     stats={k: v.cpu() for k, v in model.named_buffers() if "running" in k})
    # a rank that cannot solve its episodes: every rank raises after the collective, nobody hangs in it
    if batch:
        if rank == 0:  # the conservative schedule cannot converge in one CG iteration; rank 1 stays healthy
            model.lp_max_iter = 1
            model._lp_budget = 1
            model._lp_probe = None
        try:
            trainer.step([eps[e] for e in ids])
            out[str(rank) + "_raised"] = False
        except RuntimeError as exc:
            out[str(rank) + "_raised"] = "abandoned on all ranks" in str(exc)
    torch.distributed.destroy_process_group()



@pytest.mark.parametrize("slots,batch", [(0, 0), (2, 0), (0, 2)])
def test_two_ranks_of_dptrainer_equal_one_rank(slots, batch):
    # slots / batch: the ranks run eager launches, two captured hipGraph slots, or the episode-batched launch sequence
    # one process, all episodes, eager launches: the reference value
    cfg, model, eps, trainer = _setup(0)
    want_hist = _hist(model, eps, range(N_EPISODES)).hist.cpu()
    loss = trainer.step(eps)
    torch.cuda.synchronize()
    want_grad = trainer.bucket.flat.cpu()
    want_params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    del trainer, model
    torch.cuda.empty_cache()
    world, port = 2, _free_port()
    with mp.get_context("spawn").Manager() as mgr:  # (never fork a process that has touched the GPU: the forked manager crashed at random in garbage collection)
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, slots, out, batch), nprocs=world, join=True)
        out = dict(out)
    assert sorted(out[0]["ids"] + out[1]["ids"]) == list(range(N_EPISODES))
    assert not out[0]["redone"] and not out[1]["redone"]
    # the all-reduced bucket is the same on both ranks, bit for bit (one collective, same arithmetic after it)
    assert torch.equal(out[0]["grad"], out[1]["grad"])
    scale = want_grad.abs().max().item()
    err = (out[0]["grad"] - want_grad).abs().max().item() / scale
    # every kernel of the step sums in a fixed order: only the order of the sum over episodes differs between the runs
    assert err <= 1e-5, err
    assert torch.equal(out[0]["params"], out[1]["params"])
    perr = (out[0]["params"] - want_params).abs().max().item()
    assert perr <= 2e-3, perr  # one Adam step of lr 1e-3: sign-level agreement of the update
    assert abs(0.5 * (out[0]["loss"] + out[1]["loss"]) - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    assert torch.equal(out[0]["hist"], want_hist) and torch.equal(out[1]["hist"], want_hist)
    # running statistics: different per rank after the step, identical (their mean) after sync_running_stats()
    assert not torch.equal(out[0]["own_stat"], out[1]["own_stat"]) and out[0]["n_synced"] > 2000
    for k in out[0]["stats"]:
        assert torch.equal(out[0]["stats"][k], out[1]["stats"][k]), k
    mean = 0.5 * (out[0]["own_stat"] + out[1]["own_stat"])
    np.testing.assert_allclose(out[0]["stats"]["encoder.conv.layer.1.running_mean"].numpy(), mean.numpy(), rtol=1e-6, atol=1e-7)
    if batch:
        assert out["0_raised"] is True and out["1_raised"] is True


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg, model, eps, trainer = _setup(0, batch=2)
    loss = trainer.step(eps)
    torch.cuda.synchronize()
    out["loss"] = float(loss)
    out["grad"] = trainer.bucket.flat.cpu()
    out["backend"] = torch.distributed.get_backend()
    torch.distributed.destroy_process_group()


def test_rccl_all_reduce_executes_with_one_rank():
    """The nccl (= RCCL) backend of the training step: one rank, so the flat-bucket all-reduce, the division by the
    episode count and the failure flag run through RCCL at least once on this box (the 8-GPU node is the driver's)."""
    cfg, model, eps, trainer = _setup(0, batch=2)
    want_loss = float(trainer.step(eps))
    torch.cuda.synchronize()
    want = trainer.bucket.flat.cpu()
    del trainer, model
    torch.cuda.empty_cache()
    with mp.get_context("spawn").Manager() as mgr:  # (never fork a process that has touched the GPU: the forked manager crashed at random in garbage collection)
        out = mgr.dict()
        mp.spawn(_rccl_worker, args=(_free_port(), out), nprocs=1, join=True)
        out = dict(out)
    assert out["backend"] == "nccl"
    assert torch.equal(out["grad"], want) and abs(out["loss"] - want_loss) < 1e-6

