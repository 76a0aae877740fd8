#!/usr/bin/env python
"""Headline benchmark: episodes/sec on synthetic S3DIS-shaped 2-way 5-shot 2048-pt episodes.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one episode per rank (configs[1] of BASELINE.json:
S3DIS S0 2-way 5-shot 2048 pts, MPTI + attention), inputs resident in HBM.  Episodes are
independent: ranks shard them with no data-path collective in eval mode
(weak scaling).  Prints ONE JSON line on rank 0, with
  roofline     -- dominant kernel timed live with HIP events on its launch stream
  cpu_baseline -- the CPU oracle (a port of the reference path) on this box's host cores
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak


OPS = ["knn_topk", "knn_topk_l2", "pointwise_conv", "edgeconv", "attention", "head_prototypes", "label_propagate"]


def algorithmic_work(op, cfg, n_nodes, cg_iters):
    """ALGORITHMIC (flops, bytes) of ONE step's launches of an entry point and the roofline that bounds
    it (DESIGN.md "kernels"; per-unit figures from SURVEY.md 8d).  bytes = read every input once + write
    every output once, fp32 / int32."""
    n_way, k_shot, N = cfg["n_way"], cfg["k_shot"], cfg["pc_npts"]
    S_ = n_way * k_shot
    B = S_ + n_way * cfg.get("n_queries", 1)
    K, D = cfg["dgcnn_k"], 192
    M = B * N
    if op == "knn_topk":  # 3 launches: C = 9, 64, 64
        Cs = [cfg["pc_in_dim"], 64, 64]
        return sum(2.0 * B * N * N * C for C in Cs), sum(M * C * 4 + M * K * 4 for C in Cs), "mfma", 3
    if op == "knn_topk_l2":
        kp1 = cfg["k_connect"] + 1
        return 2.0 * n_nodes * n_nodes * D, n_nodes * D * 4 + n_nodes * kp1 * 4, "mfma", 1
    if op == "edgeconv":  # 3 launches
        return 3 * M * K * (2.0 * 64 * 64 + 3 * 64), 3 * (M * 128 * 4 + M * K * 4 + M * 64 * 4), "mfma", 3
    if op == "pointwise_conv":  # PQ x3, mlp x2, base x2, qkv
        shapes = [(cfg["pc_in_dim"], 128), (64, 128), (64, 128), (192, 512), (512, 256), (256, 128), (128, 64), (256, 192)]
        return (sum(2.0 * M * k * co for k, co in shapes), sum(M * k * 4 + M * co * 4 + k * co * 4 for k, co in shapes),
                "mfma", len(shapes))
    if op == "attention":
        return 4.0 * B * N * N * 64, M * 192 * 4 + M * 64 * 4, "mfma", 1
    if op == "head_prototypes":  # FPS + assignment + means: features read once per pass
        pts = S_ * N
        ksub = cfg["n_subprototypes"]
        return 3.0 * ksub * pts * D * 2, 3 * pts * D * 4, "hbm", 1
    if op == "label_propagate":  # bitmap + CSR build + cg_iters SpMVs over <= 2*k nnz per row
        nnz = 2.0 * n_nodes * cfg["k_connect"]
        return (n_nodes * cfg["k_connect"] * D * 6 + cg_iters * nnz * 8,
                n_nodes * D * 4 + nnz * 8 + cg_iters * (nnz * 8 + n_nodes * 16 * 4), "hbm", 1)
    raise KeyError(op)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="S", choices=["S", "C", "P"])
    ap.add_argument("--mode", default="eval", choices=["eval", "train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-kernel", default="auto", help="entry point to price (auto = the one taking most time)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from r3dfsseg_amd import ops, synthetic as S
    from r3dfsseg_amd.mpti import MPTI_SelfAtten

    cfg = S.workload_cfg(args.workload)
    model = MPTI_SelfAtten(SimpleNamespace(**cfg))
    model.load_state_dict(S.make_state_dict(cfg, 123))
    model.to(dev)
    train = args.mode == "train"
    n_pool = 8  # distinct episodes per rank, resident in HBM before timing starts
    pool = []
    for e in range(n_pool):
        data, _ = S.make_episode(cfg, seed=1000 * rank + e, noise_ratio=0.2 if train else 0.0, train=train)
        pool.append([t.to(dev) for t in (data if train else data[:4])])
    torch.cuda.synchronize()

    lp_flags = []
    if train:
        # forward + backward + single flat-bucket gradient all-reduce + Adam (mpti_learner.py:60-72)
        from r3dfsseg_amd.dp_train import DPTrainer
        targs = SimpleNamespace(lr=1e-3, step_size=5000, gamma=0.5, **cfg)
        learner = SimpleNamespace(model=model)
        learner.optimizer = torch.optim.Adam(
            [{'params': model.encoder.parameters(), 'lr': 0.0001}, {'params': model.base_learner.parameters()},
             {'params': model.att_learner.parameters()}, {'params': model.proj.parameters()}], lr=targs.lr)
        learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=targs.step_size, gamma=targs.gamma)
        trainer = DPTrainer(learner)
        model.train()

        def step(i):
            loss = trainer.step([pool[i % n_pool]])
            hb = model._head[1]
            lp_flags.append(torch.cat((hb.stats * hb.stats_bwd[:1].clamp(max=1), hb.knn_status)))
            return loss
    else:
        model.eval()

        def step(i):
            sx, sy, qx, qy = pool[i % n_pool]
            with torch.no_grad():
                logits, loss = model(sx, sy, qx, qy)
            hb = model._head[1]
            lp_flags.append(torch.cat((hb.stats, hb.knn_status)))  # (CG converged, CG iterations, kNN overflow)
            return logits, loss

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    # roofline leg: the same K steps again with HIP events around every entry point (events are
    # recorded on the launch stream = torch's current stream); kept out of the timed region above
    timer = ops.KernelTimer(OPS)
    ops.set_timer(timer)
    for i in range(args.steps):
        step(i)
    ops.set_timer(None)
    ksum_all = timer.summary()
    lp = torch.stack(lp_flags[args.warmup:args.warmup + args.steps]).cpu()
    if int(lp[:, 2].max()) != 0:
        raise SystemExit("bench invalid: 201-NN survivor buffer overflowed")
    if int(lp[:, 0].min()) != 1:
        raise SystemExit("bench invalid: label propagation did not converge in %d timed episode(s)" % int((lp[:, 0] != 1).sum()))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    S_ = cfg["n_way"] * cfg["k_shot"]
    B = S_ + cfg["n_way"] * cfg.get("n_queries", 1)
    N = cfg["pc_npts"]
    n_nodes = int(model._head[1].desc[ops.HD_N_NODES].item())
    cg_mean = float(lp[:, 1].float().mean())
    per_step_ms = {k: v["total_ms"] / args.steps for k, v in ksum_all.items()}
    kern = max(per_step_ms, key=per_step_ms.get) if args.roofline_kernel == "auto" else args.roofline_kernel
    fl, by, bound, launches = algorithmic_work(kern, cfg, n_nodes, cg_mean)
    t_launch = per_step_ms[kern] * 1e-3 / launches
    if bound == "mfma":
        ach, peak, unit = fl / launches / t_launch / 1e12, F32_MFMA_PEAK_TF, "TFLOP/s"
    else:
        ach, peak, unit = by / launches / t_launch / 1e9, HBM_PEAK_GBS, "GB/s"
    roof = dict(kernel=kern, bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, traffic=None,
                avg_launch_ms=t_launch * 1e3, launches_per_step=launches,
                algorithmic_gflop_per_launch=fl / launches / 1e9, algorithmic_mb_per_launch=by / launches / 1e6,
                hbm_gbs=by / launches / t_launch / 1e9, fp32_tflops=fl / launches / t_launch / 1e12)
    breakdown = {k: round(v, 4) for k, v in sorted(per_step_ms.items(), key=lambda kv: -kv[1])}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        # the GPU box hands one GPU a 16-core CPU share (os.cpu_count() reports the whole host)
        ncores = min(os.cpu_count() or 1, int(os.environ.get("R3D_CPU_THREADS", "16")))
        os.environ["OMP_NUM_THREADS"] = str(ncores)
        from oracle import r3d_oracle as O
        torch.set_num_threads(ncores)
        sd = S.make_state_dict(cfg, 123)
        data, _ = S.make_episode(cfg, seed=0)
        sx, sy, qx, qy = data[:4]
        O.knn(qx[:1, :, :256], 4)  # load + warm the C library
        c0 = time.perf_counter()
        n_cpu = 0
        while True:
            O.mpti_forward(sd, cfg, sx, sy, qx, qy)
            n_cpu += 1
            if time.perf_counter() - c0 > 10.0 or n_cpu >= 3:
                break
        c1 = time.perf_counter()
        cpu = dict(value=n_cpu / (c1 - c0), unit="episodes/s", cores=ncores, kind="port",
                   sample="%d full %s episode(s), eval forward, CPU oracle (C + torch-CPU, %d threads)" % (n_cpu, args.workload, ncores))

    eps = args.steps * world / elapsed
    out = {
        "metric": "episodes/sec S3DIS 2-way 5-shot 2048-pt (MPTI+attention, %s)" % (
            "train: forward+backward+grad all-reduce+Adam" if train else "eval forward"),
        "value": eps, "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d-way %d-shot %d pts, %d clouds/episode, 1 episode/step/rank, mode=%s" % (
            args.workload, cfg["n_way"], cfg["k_shot"], N, B, args.mode), "episodes_per_step": world},
        "roofline": roof, "cpu_baseline": cpu,
        "entry_point_ms_per_step": breakdown,
        "lp_cg_iterations": {"mean": float(lp[:, 1].float().mean()), "max": int(lp[:, 1].max())},
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
