"""Throughput of the episode-batched training step / eval forward at a workload, for a list of batch sizes.
    python tools/batch_bench.py [--workload S] [--episodes 32] [--batch 8,16,32] [--steps 10] [--mode train|eval]"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import synthetic as S  # noqa: E402
from r3dfsseg_amd.batch import EpisodeBatch  # noqa: E402
from r3dfsseg_amd.mpti import MPTI_SelfAtten  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S")
    ap.add_argument("--episodes", type=int, default=32)
    ap.add_argument("--batch", default="8,16,32")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="train")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = S.workload_cfg(args.workload)
    E = args.episodes
    eps = []
    for e in range(E):
        data, _ = S.make_episode(cfg, seed=e, noise_ratio=0.2, train=True)
        eps.append([t.to(dev) for t in data])
    for B in [int(b) for b in args.batch.split(",")]:
        model = MPTI_SelfAtten(SimpleNamespace(**cfg))
        model.load_state_dict(S.make_state_dict(cfg, 123))
        model.to(dev)
        batches = [EpisodeBatch.from_episodes(eps[i:i + B]) for i in range(0, E, B)]
        if args.mode == "train":
            from r3dfsseg_amd.dp_train import DPTrainer
            learner = SimpleNamespace(model=model)
            learner.optimizer = torch.optim.Adam(
                [{'params': model.encoder.parameters(), 'lr': 0.0001}, {'params': model.base_learner.parameters()},
                 {'params': model.att_learner.parameters()}, {'params': model.proj.parameters()}], lr=1e-3)
            learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
            model.train()
            tr = DPTrainer(learner, batch_size=B)
            step = lambda: tr.step(batches)
        else:
            from r3dfsseg_amd.batched import EpisodeBatchRunner
            model.eval()
            run = EpisodeBatchRunner(model)

            def step():
                run.begin_step()
                for b in batches:
                    run.eval_batch(b)
                return run.step_status()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        extra = ""
        if args.mode == "train":
            extra = " redone %d status %s" % (tr.n_redone, tr.last_status)
        print("%s %s: batch %3d  %.1f episodes/s  (%.2f ms/step, peak mem %.1f GB)%s" % (
            args.mode, args.workload, B, args.steps * E / el, el / args.steps * 1e3,
            torch.cuda.max_memory_allocated() / 2**30, extra), flush=True)
        del model
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
