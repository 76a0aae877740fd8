"""Evaluation sweep straight from cached `.h5` episode files -- the reference's on-disk container (dataloaders/loader.py:
1687-1721) read through episode_io (HDF5 C library), collated as point-major views, 32 episodes per launch sequence --
against the same sweep on episodes already resident in HBM: what the disk read, the host-to-device copy and the collate
cost end to end.   usage: eval_from_cache.py [episodes=128] [workload=S]"""
import os, sys, tempfile, time, shutil
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from r3dfsseg_amd import episode_io as EIO, synthetic as S
from r3dfsseg_amd.batch import EpisodeBatch
from r3dfsseg_amd.batched import EpisodeBatchRunner
from r3dfsseg_amd.metrics import MIoUAccumulator
from r3dfsseg_amd.mpti import MPTI_SelfAtten

n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = sys.argv[2] if len(sys.argv) > 2 else "S"
E = 32
cfg = S.workload_cfg(W)
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().eval()
root = tempfile.mkdtemp(prefix="r3d_cache_")
try:
    t0 = time.time()
    for i in range(n_ep):
        data, classes = S.make_episode(cfg, seed=2000 + i, noise_ratio=0.4)
        sx, sy, qx, qy = data[:4]
        raw = (sx.transpose(2, 3).contiguous().numpy(), sy.numpy(), qx.transpose(1, 2).contiguous().numpy(), qy.numpy(),
               np.asarray(classes), np.zeros(sy.shape, np.int32), np.zeros(qy.shape, np.int32), data[6].numpy())
        EIO.write_episode(os.path.join(root, "%d.h5" % i), raw)
    files = EIO.list_episode_files(root)
    size = sum(os.path.getsize(f) for f in files)
    print("wrote %d .h5 episodes, %.1f MB, in %.1f s" % (n_ep, size / 1e6, time.time() - t0), flush=True)
    run = EpisodeBatchRunner(m)
    test_classes = sorted({int(c) for i in range(n_ep) for c in S.make_episode(cfg, seed=2000 + i)[1]})

    def sweep(source):
        acc = MIoUAccumulator(test_classes)
        run.begin_step()
        group, cls = [], []
        done = 0
        for out, classes in source:
            group.append(out); cls.append(classes)
            if len(group) == E:
                b = EpisodeBatch.from_episodes(group)
                logits, _ = run.eval_batch(b)
                pred = logits.argmax(2)
                for e in range(E):
                    acc.update(pred[e], b.query_y[e], cls[e])
                done += E
                group, cls = [], []
        bad, ovf, _, _ = run.step_status()
        assert not bad and not ovf, (bad, ovf)
        return done, acc.compute()[0]

    sweep(EIO.EpisodeFeeder(files[:E], device="cuda"))  # warm-up (kernels, allocator)
    torch.cuda.synchronize(); t0 = time.time()
    done, miou = sweep(EIO.EpisodeFeeder(files, device="cuda", depth=64))
    torch.cuda.synchronize(); t_disk = time.time() - t0
    resident = [EIO.collate_test(EIO.read_episode(f), "cuda") for f in files]
    torch.cuda.synchronize(); t0 = time.time()
    done2, miou2 = sweep(iter(resident))
    torch.cuda.synchronize(); t_res = time.time() - t0
    assert done == done2 and miou == miou2
    print("eval sweep of %d episodes (32 per launch sequence): from .h5 files %.1f episodes/s, resident in HBM %.1f episodes/s; "
          "mIoU %.4f both ways" % (done, done / t_disk, done / t_res, miou))
finally:
    shutil.rmtree(root)
