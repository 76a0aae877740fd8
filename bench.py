#!/usr/bin/env python
"""Headline benchmark: episodes/sec on synthetic S3DIS-shaped 2-way 5-shot 2048-pt episodes.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of E episodes per rank (--episodes-per-rank, default 32 =
BASELINE.json configs[4]'s 256-episode batch over 8 GPUs; each episode is configs[1]: S3DIS S0 2-way 5-shot
2048 pts, MPTI + attention), inputs resident in HBM.
  --mode train (default): E x (forward + backward) + ONE flat-bucket RCCL gradient all-reduce + Adam.  With
                          E = 1 this is the reference's MPTILearner_V3.train step (models/mpti_learner.py:60-72).
  --mode eval           : forward only, MPTILearner_V3.test without its host sync.
  --batch B (default E) : episodes per launch sequence -- the E episodes of a step go through the kernels TOGETHER
                          (r3dfsseg_amd/batched.py: one launch sequence per batch of B episodes); --batch 0 = eager
                          launches, one episode at a time (the reference's schedule).
Episodes are independent: ranks take disjoint episodes (weak scaling); the only collective is
the 1.5 MB gradient all-reduce of train mode.  ONE JSON line on rank 0.
  value        -- STEADY STATE: K timed steps after --steady-steps optimiser steps from the synthetic initial weights
                  (the label-propagation systems get harder as the encoder separates the classes: CG iterations
                  roughly double within 150 steps; the reference trains 40 000 episodes, mpti_train_noise.py:182).
                  `fresh_weights` is the same measurement over the first K steps after W warm-up steps.
  roofline     -- the kernel with the most device time in a live batched step, timed with HIP events on the launch
                  stream (one event pair per entry-point call; batched launches do not overlap) and priced with its
                  ALGORITHMIC work.  When that entry point is the label propagation, the CG
                  iteration (its two kernels, all systems of the batch per launch) is measured on its own.
  rooflines    -- every entry point of the batched step against both ceilings, same timing
  cpu_baseline -- the CPU oracle (a port of the reference path) on this box's host cores: EVAL FORWARD (1 warm-up,
                  min / median of 5) to be compared with eval_forward_episodes_per_sec, and a TRAIN STEP (forward,
                  backward through the dense inverse, Adam; 1 warm-up, median of 3) to be compared with
                  single_episode_eager_step_episodes_per_sec (same schedule) or `value`
Extra fields: single_episode_eager_step_episodes_per_sec (E = 1, eager: the reference's own schedule),
single_episode_graph_step_episodes_per_sec (the same as one captured hipGraph) and eval_forward_episodes_per_sec.
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
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)

OPS = ["knn_topk", "knn_topk_l2", "pointwise_conv", "edgeconv", "attention", "head_prototypes", "label_propagate",
       "gemm_tn", "edgeconv_bwd", "attention_bwd", "bn_stats", "label_propagate_bwd"]
FETCH_SIZE_WIDE_READ_FACTOR = 2.0  # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half of wide coalesced reads


def algorithmic_work(op, cfg, n_nodes, cg_iters, train):
    """ALGORITHMIC (flops, bytes, bound, launches) of ONE EPISODE's share of an entry point (DESIGN.md section 4;
    per-unit figures from SURVEY.md 8d); a batched step's launches carry E times this.  bytes = every input read once
    + every output written once (fp32 / int32)."""
    n_way, k_shot, N = cfg["n_way"], cfg["k_shot"], cfg["pc_npts"]
    S_ = n_way * k_shot
    n_q = n_way * cfg.get("n_queries", 1)
    K, D = cfg["dgcnn_k"], 192
    M = (S_ + n_q) * N
    passes = 1   # (BatchNorm keeps the getFeatures calls apart as segments inside one launch)
    joint = 1  # kNN, GEMMs, attention and the edge passes run once over all clouds
    conv_shapes = [(cfg["pc_in_dim"], 128), (64, 128), (64, 128), (192, 512), (512, 256), (256, 128), (128, 64), (256, 192)]
    if op == "knn_topk":  # per pass 3 launches: C = 9, 64, 64; sum over clouds of 2 N^2 C
        Cs = [cfg["pc_in_dim"], 64, 64]
        return sum(2.0 * (S_ + n_q) * N * N * C for C in Cs), sum(M * C * 4 + M * K * 4 for C in Cs), "mfma", 3 * joint
    if op == "knn_topk_l2":
        kp1 = cfg["k_connect"] + 1
        return 2.0 * n_nodes * n_nodes * D, n_nodes * D * 4 + n_nodes * kp1 * 4, "mfma", 1
    if op == "edgeconv":  # eval: 3 fused launches; train: per layer BN1 statistics, ONE edge-GEMM pass (min/max), select
        f = 3 * M * K * (2.0 * 64 * 64 + 3 * 64)
        b = 3 * (M * 128 * 4 + M * K * 4 + M * 64 * 4)
        bt = 3 * (2 * M * 128 * 4 + 2 * M * K * 4 + 9 * M * 64 * 4)
        return (f, bt, "mfma", 9 * passes) if train else (f, b, "mfma", 3)
    if op == "pointwise_conv":
        f = sum(2.0 * M * k * co for k, co in conv_shapes)
        b = sum(M * k * 4 + M * co * 4 + k * co * 4 for k, co in conv_shapes)
        return (f * 2, b * 2, "mfma", 2 * len(conv_shapes) * joint) if train else (f, b, "mfma", len(conv_shapes))
    if op == "attention":
        return 4.0 * (S_ + n_q) * N * N * 64, M * 192 * 4 + M * 64 * 4, "mfma", joint
    if op == "attention_bwd":  # S recomputed twice, dP twice, dV, dK, dQ: 14 N^2 d per cloud
        return 14.0 * (S_ + n_q) * N * N * 64, M * (192 + 64 + 64 + 192) * 4, "mfma", joint
    if op == "gemm_tn":  # weight gradients of every conv: 2 M K Co each
        shapes = conv_shapes[:]
        return (sum(2.0 * M * k * co for k, co in shapes), sum(M * (k + co) * 4 + k * co * 4 for k, co in shapes), "mfma",
                len(shapes) * joint)
    if op == "edgeconv_bwd":  # three edge GEMMs (z2 recompute, dh1, dW2) + dy1 round trip
        return 3 * M * K * (3 * 2.0 * 64 * 64), 3 * (M * 128 * 4 * 2 + 2 * M * K * 64 * 4 + M * 64 * 8), "mfma", 3 * passes
    if op == "edgeconv_bwd_useful":  # what the math needs: dh1 = dz2 W2 and dW2 += dz2^T h1 (the z2 GEMM is a recompute)
        return 3 * M * K * (2 * 2.0 * 64 * 64), 3 * (M * 128 * 4 * 2 + 2 * M * K * 64 * 4 + M * 64 * 8), "mfma", 3 * passes
    if op == "bn_stats":  # column statistics: every activation / gradient matrix read once
        cols = [512, 256, 128, 64]
        return 4.0 * M * sum(cols) * 3, 3 * M * sum(cols) * 4, "hbm", 3 * len(cols) * passes
    if op == "head_prototypes":  # FPS + assignment + means: features read once per pass
        pts = S_ * N
        return 3.0 * cfg["n_subprototypes"] * pts * D * 2, 3 * pts * D * 4, "hbm", 1
    if op in ("label_propagate", "label_propagate_bwd"):  # graph build + cg_iters SpMVs over <= 2 k nnz per row
        nnz = 2.0 * n_nodes * cfg["k_connect"]
        return (n_nodes * cfg["k_connect"] * D * 6 + cg_iters * nnz * 8,
                n_nodes * D * 4 + nnz * 6 + cg_iters * cg_iteration_bytes(nnz, n_nodes), "hbm", 1)
    raise KeyError(op)


CG_M = 64  # coarse dimensions of the two-level CG (csrc/head_graph.hip: HG_M)


def cg_iteration_bytes(nnz, n):
    """ALGORITHMIC bytes of ONE CG iteration (kernels R + S + U of csrc/head_graph.hip): the matrix once (uint16 column
    + fp32 value per entry), the gathered residual rows counted once, the vectors each kernel reads and writes
    (S: r in, p and q in/out, one (M W) row; U: p, q in, x and r in/out, one (M W) row), float4 rows of 16 B."""
    return nnz * 6.0 + n * (16 + 32 + 32 + 4 * CG_M) + n * (32 + 32 + 32 + 4 * CG_M)


# entry point -> the kernel that dominates it (names as rocprofv3 prints them; profiles/)
MAIN_KERNEL = {
    "knn_topk": "r3d_knn_append_kernel<4, 128, 1, ...> (DGCNN kNN, k = 20)",
    "knn_topk_l2": "r3d_knn_append_kernel<8, 384, 2, ...> (201-NN of the graph nodes)",
    "pointwise_conv": "r3d_pointwise_gemm_bx3_kernel (bf16 x 3; fp32 arithmetic: r3d_pointwise_gemm_kernel)", "edgeconv": "r3d_edgeconv_kernel / r3d_edgeconv_train_fwd2_kernel",
    "attention": "r3d_attention_fwd_bx3_kernel (bf16 x 3; fp32 arithmetic: r3d_attention_fwd_kernel)",
    "head_prototypes": "r3d_fps_persistent_kernel",
    "label_propagate": "r3d_cg_spmv_lds_kernel + r3d_cg_update_kernel",
    "label_propagate_bwd": "r3d_cg_spmv_lds_kernel + r3d_cg_update_kernel",
    "gemm_tn": "r3d_gemm_tn_bx3_kernel (bf16 x 3; fp32 arithmetic: r3d_gemm_tn_kernel)", "edgeconv_bwd": "r3d_edgeconv_bwd1_bx3_kernel + r3d_edgeconv_bwd2_kernel (bf16 x 3; fp32 arithmetic: r3d_edgeconv_bwd1_kernel)",
    "attention_bwd": "r3d_attention_bwd_kv_bx3_kernel + r3d_attention_bwd_q_bx3_kernel (fp32 arithmetic: ..._kv_kernel + ..._q_kernel)",
    "bn_stats": "r3d_colpartial_kernel",
}


BX3_OPS = ("attention", "attention_bwd", "pointwise_conv", "gemm_tn", "edgeconv_bwd")


def _lib_arith():
    from r3dfsseg_amd import _lib
    return _lib.load().r3d_get_matrix_arith()


def committed_profile(workload, mode):
    """({kernel name as printed: (calls, total ms, average us)}, file name) of the newest committed rocprofv3
    --kernel-trace summary of THIS workload and mode
    (profiles/rNN_*_rocprofv3_kernel_stats_<mode>_eager_<workload>.txt, written by tools/prof_summary.py), or (None, None)."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats_%s_batched_%s.txt" % (mode, workload))))
    if not names:
        return None, None
    rows = {}
    for l in open(names[-1]):
        if l.startswith("#"):  # "# rocprofv3 --kernel-trace -- python3 tools/one_step.py MODE W E STEPS   (...": steps of the profile
            try:
                rows["_steps"] = int(l.split("(")[0].split()[-1])
            except (ValueError, IndexError):
                pass
            continue
        f = l.split()
        if len(f) >= 5 and "r3d_" in l[:16]:
            try:
                rows[l[:72].strip()] = (int(f[-4]), float(f[-3]), float(f[-2]))
            except ValueError:
                pass
    return rows, os.path.basename(names[-1])


def committed_traffic(workload, mode):
    """per-kernel {fetch_kb_per_launch, write_kb_per_launch} of the committed PMC passes of THIS workload, or None."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % workload)))
    if not names:
        return None, None
    try:
        return json.load(open(names[-1]))[mode], os.path.basename(names[-1])
    except (OSError, KeyError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="S", choices=["S", "C", "P"])
    ap.add_argument("--mode", default="train", choices=["eval", "train"])
    ap.add_argument("--episodes-per-rank", type=int, default=32,
                    help="episodes of one step on every rank (BASELINE configs[4]: 256-episode batch / 8 GPUs)")
    ap.add_argument("--batch", type=int, default=None,
                    help="episodes per launch sequence (default: all of the rank's episodes; 0 = one episode at a time, eager)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-episode and eval legs (profiling runs)")
    ap.add_argument("--no-batch-graph", action="store_true",
                    help="launch the batched training step eagerly instead of replaying its captured hipGraph (A/B; same results)")
    ap.add_argument("--roofline-kernel", default="auto", help="entry point to price (auto = the one taking most time)")
    ap.add_argument("--steady-steps", type=int, default=150,
                    help="train mode: optimiser steps from the initial weights before the headline is timed (0 = headline on fresh weights)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    if os.environ.get("R3D_BENCH_ONE_DEVICE"):  # rehearsal of the N > 1 code path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("R3D_BENCH_FORCE_DIST"):  # FORCE_DIST: rehearse the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from r3dfsseg_amd import ops, synthetic as S
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.batched import EpisodeBatchRunner
    from r3dfsseg_amd.dp_train import DPTrainer
    from r3dfsseg_amd.mpti import MPTI_SelfAtten

    cfg = S.workload_cfg(args.workload)
    model = MPTI_SelfAtten(SimpleNamespace(**cfg))
    model.load_state_dict(S.make_state_dict(cfg, 123))
    model.to(dev)

    E = args.episodes_per_rank
    Bsz = E if args.batch is None else args.batch
    batched = Bsz > 0
    # the synthetic pool: 2 E distinct episodes per rank, resident in HBM (and already collated into batches of Bsz
    # episodes) before any timing starts; consecutive steps alternate between the two halves
    n_pool = 2 * E
    pool = []
    for e in range(n_pool):
        data, _ = S.make_episode(cfg, seed=1000 * rank + e, noise_ratio=0.2, train=True)
        pool.append([t.to(dev) for t in data])
    halves = [pool[:E], pool[E:]]
    pool_batches = [[EpisodeBatch.from_episodes(h[i:i + Bsz]) for i in range(0, E, Bsz)] for h in halves] if batched else None
    torch.cuda.synchronize()

    learner = SimpleNamespace(model=model)
    learner.optimizer = torch.optim.Adam(
        [{'params': model.encoder.parameters(), 'lr': 0.0001}, {'params': model.base_learner.parameters()},
         {'params': model.att_learner.parameters()}, {'params': model.proj.parameters()}], lr=1e-3)
    learner.lr_scheduler = torch.optim.lr_scheduler.StepLR(learner.optimizer, step_size=5000, gamma=0.5)
    train = args.mode == "train"
    model.train(train)
    trainer = DPTrainer(learner, batch_size=Bsz, batch_graph=not args.no_batch_graph) if train else None
    runner = trainer.runner if (train and batched) else (EpisodeBatchRunner(model) if batched else None)
    lp_flags = []
    cg_acc = [0, 0, 0]    # CG iterations of the forward solves: sum, systems, max
    invalid = []          # reasons why a timed leg does not count (every episode must have a converged, exact head)
    redone_notes = []

    def head_flags():
        hb = model._head[1]
        return torch.cat((hb.stats.view(-1)[:2], hb.knn_status))  # (CG converged, CG iterations, kNN overflow)

    def train_step(i):  # E episodes through the batched launch sequences -> ONE all-reduce -> Adam
        trainer.step(pool_batches[i & 1] if batched else halves[i & 1])
        if batched:
            bad, ovf, its, mx = trainer.last_status
            cg_acc[0] += its; cg_acc[1] += E; cg_acc[2] = max(cg_acc[2], mx)

    def train_step_eager(i):  # the reference's schedule: one episode, eager launches, all-reduce, Adam
        saved = trainer.runner, trainer.graphs
        trainer.runner, trainer.graphs = None, None
        trainer.step([pool[i % n_pool]])
        trainer.runner, trainer.graphs = saved
        hb = model._head[1]
        f = head_flags()
        f[0] = f[0] * hb.stats_bwd.view(-1)[0].clamp(max=1)  # forward AND adjoint solve converged
        lp_flags.append(f)

    def eval_step(i):
        model.eval()
        runner.begin_step()
        for b in pool_batches[i & 1]:
            runner.eval_batch(b)
        bad, ovf, its, mx = runner.step_status()
        cg_acc[0] += its; cg_acc[1] += E; cg_acc[2] = max(cg_acc[2], mx)
        if bad or ovf:
            invalid.append("eval step %d: %d system(s) did not converge / timed out, %d batch(es) with 201-NN overflow" % (i, bad, ovf))

    def eval_step_eager(i):
        model.eval()
        sx, sy, qx, qy = pool[i % n_pool][:4]
        with torch.no_grad():
            model(sx, sy, qx, qy)
        lp_flags.append(head_flags())

    def timed(step_fn, steps, warmup):
        for i in range(warmup):
            step_fn(i)
        del lp_flags[:]
        cg_acc[:] = [0, 0, 0]
        redone0 = trainer.n_redone if trainer is not None else 0
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(warmup + i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        if trainer is not None and trainer.n_redone > redone0:  # DPTrainer.step fails closed: redone exactly, inside the timed region
            redone_notes.append("%s: %d step(s) redone on the conservative schedule" % (step_fn.__name__, trainer.n_redone - redone0))
        if lp_flags:
            lp = torch.stack(lp_flags).cpu()
            if int(lp[:, 2].max()) != 0:
                invalid.append("%s: 201-NN survivor buffer overflowed" % step_fn.__name__)
            if int(lp[:, 0].min()) != 1:
                invalid.append("%s: label propagation did not converge in %d episode(s)" % (step_fn.__name__, int((lp[:, 0] != 1).sum())))
            return el, (float(lp[:, 1].float().mean()), int(lp[:, 1].max()))
        return el, (cg_acc[0] / max(cg_acc[1], 1), cg_acc[2])

    extra = {}
    steps_done = 0
    if train:
        main_step = train_step if batched else train_step_eager
        el_f, cg_f = timed(main_step, args.steps, args.warmup)
        steps_done = args.steps + args.warmup
        Eeff = E if batched else 1
        fresh = {"value": args.steps * Eeff * world / el_f, "unit": "episodes/s", "ms_per_step": el_f / args.steps * 1e3,
                 "lp_cg_iterations": {"mean": cg_f[0], "max": cg_f[1]}, "optimiser_steps_before": args.warmup}
        if batched and args.steady_steps > steps_done:
            for i in range(args.steady_steps - steps_done):  # conditioning: train on (all ranks, untimed)
                train_step(steps_done + i)
            steps_done = args.steady_steps
            elapsed, cg = timed(train_step, args.steps, 0)
            steps_done += args.steps
            extra["fresh_weights"] = fresh
        else:
            elapsed, cg = el_f, cg_f
        E = Eeff
        if batched and not args.no_extras:
            n_e = max(args.steps, 16)
            el1, _ = timed(train_step_eager, n_e, 3)
            extra["single_episode_eager_step_episodes_per_sec"] = n_e * world / el1
            saved_runner = trainer.runner
            try:  # the same schedule as ONE captured hipGraph per episode (episode_graph.py, MPTILearner_V3's episode_graphs switch)
                from r3dfsseg_amd.episode_graph import EpisodeGraphs
                rows = torch.zeros(1, trainer.bucket.store.numel(), device=dev)
                trainer.graphs = EpisodeGraphs(model, pool[0], 1, train=True, grad_rows=rows)
                trainer.rows = rows
                trainer.runner = None

                def train_step_single_graph(i):
                    trainer.step([pool[i % n_pool]])
                el1g, _ = timed(train_step_single_graph, n_e, 3)
                extra["single_episode_graph_step_episodes_per_sec"] = n_e * world / el1g
            except Exception as exc:  # an extra, never the headline
                extra["single_episode_graph_step_error"] = repr(exc)[:200]
            trainer.runner, trainer.graphs = saved_runner, None
            n_ev = max(args.steps // 2, 2)
            el_ev, cg_ev = timed(eval_step, n_ev, 2)
            extra["eval_forward_episodes_per_sec"] = n_ev * E * world / el_ev
            extra["eval_forward_lp_cg_iterations"] = {"mean": cg_ev[0], "max": cg_ev[1]}
            model.train()
    else:
        if batched:
            elapsed, cg = timed(eval_step, args.steps, args.warmup)
        else:
            E = 1
            elapsed, cg = timed(eval_step_eager, args.steps, args.warmup)
    cg_mean, cg_max = cg

    # roofline leg: ONE more step of the headline kind with a HIP event pair (torch's current stream = the launch stream)
    # around every entry-point call; batched launches run one after the other, so the pairs do not overlap
    step_fn = (train_step if batched else train_step_eager) if train else (eval_step if batched else eval_step_eager)
    n_roof = 2
    timer = ops.KernelTimer(OPS, repeat=0)
    ops.set_timer(timer)
    if trainer is not None:
        graph_mode, trainer.batch_graph = trainer.batch_graph, False  # the event pairs sit around eager entry-point calls
        ops.set_timer(None)
        step_fn(0)  # (the eager launch path's first step after the graph replays: allocations, lazily set attributes)
        ops.set_timer(timer)
    for i in range(n_roof):
        step_fn(i)
    if trainer is not None:
        trainer.batch_graph = graph_mode
    ops.set_timer(None)
    timer.close()
    ksum_all = timer.summary()

    n_invalid = torch.tensor([float(len(invalid))], device=dev)
    if dist is not None:
        dist.all_reduce(n_invalid)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    N = cfg["pc_npts"]
    B = cfg["n_way"] * cfg["k_shot"] + cfg["n_way"] * cfg.get("n_queries", 1)
    hb = model._head[1]
    n_nodes = int(hb.desc.view(-1, 32)[0, ops.HD_N_NODES].item())
    per_step_ms = {k: v["total_ms"] / n_roof for k, v in ksum_all.items() if v["launches"]}
    calls_per_step = {k: v["launches"] / n_roof for k, v in ksum_all.items() if v["launches"]}
    mode_name = "train" if train else "eval"
    prof, prof_name = committed_profile(args.workload, mode_name)
    pmc, pmc_name = committed_traffic(args.workload, mode_name)
    kern = args.roofline_kernel
    if kern == "auto":  # the entry point with the most device time in the live step above
        kern = max(per_step_ms, key=per_step_ms.get)
    roof = None
    cg_roof = None
    if True:
        # The label propagation is graph build + CG; its dominant kernels are the two of one CG iteration, which serve
        # every system of the batch.  Their time is measured live: the same solves with 8 and with 40 forced iterations
        # (tol = 0), HIP events on the launch stream, difference / 32.
        nbr = ops.knn_nodes(hb)

        def solve_ms(iters, reps=4):
            ops.label_propagate(hb, nbr, model.sigma, 0.99, iters, 0.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                ops.label_propagate(hb, nbr, model.sigma, 0.99, iters, 0.0)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        t_iter = (solve_ms(40) - solve_ms(8)) / 32.0 * 1e-3
        nnz = sum(int(hb.csr(e)[1][-1].item()) for e in range(hb.E))
        nodes_tot = int(hb.desc.view(-1, 32)[:, ops.HD_N_NODES].sum().item())
        by = cg_iteration_bytes(nnz, nodes_tot)
        fl = nnz * 2.0 * 4 + nodes_tot * 4 * (12.0 + 4 * CG_M)
        # (batches of >= 8 systems run the LDS-resident SpMV, smaller ones the row-per-wave form: same bits)
        cg_names = ("r3d_cg_spmv_lds_kernel" if hb.E * ((hb.n_cap + 127) // 128) >= 256 else "r3d_cg_spmv_kernel",
                    "r3d_cg_update_kernel")
        traffic = rocprof_us = None
        if pmc is not None and all(k in pmc for k in cg_names):
            traffic = 1024.0 * sum(FETCH_SIZE_WIDE_READ_FACTOR * pmc[k]["fetch_kb_per_launch"] + pmc[k]["write_kb_per_launch"]
                                   for k in cg_names)
        if prof is not None:
            us = [v[2] for k, v in prof.items() if k.split("(")[0] in cg_names]
            rocprof_us = sum(us) if len(us) == 2 else None
        cg_roof = dict(kernel="%s + %s (one iteration of the two-level CG of the label propagation, %d systems per launch, "
                           "graphs of the weights at the time of measurement: hub rows grow with training; entry points "
                           "label_propagate / label_propagate_bwd)" % (cg_names + (hb.E,)),
                    bound="hbm", achieved=by / t_iter / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=by / t_iter / 1e9 / HBM_PEAK_GBS,
                    traffic=traffic, avg_launch_ms=t_iter * 1e3, algorithmic_mb_per_launch=by / 1e6,
                    algorithmic_gflop_per_launch=fl / 1e9, csr_nnz=nnz, nodes=nodes_tot, systems_per_launch=hb.E,
                    iterations_per_episode=cg_mean * (2 if train else 1), rocprofv3_kernel_us=rocprof_us,
                    frac_kernel_time_only=(by / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if rocprof_us else None,
                    rocprofv3_summary=prof_name, pmc_summary=pmc_name,
                    note=("avg_launch_ms is event-measured over back-to-back launches (pair of kernels); rocprofv3_kernel_us is "
                          "the pair's in-kernel time from the committed --kernel-trace summary of this workload (null when none "
                          "is committed); traffic = %.0f x FETCH_SIZE + WRITE_SIZE of the pair from the committed PMC passes "
                          "(the guide's gfx950 correction for wide coalesced reads)" % FETCH_SIZE_WIDE_READ_FACTOR))
    if kern in ("label_propagate", "label_propagate_bwd"):
        roof = cg_roof
    if roof is None:
        fl, by, bound, launches = algorithmic_work(kern, cfg, n_nodes, cg_mean, train)
        fl, by = fl * E, by * E                       # the step's work of this entry point
        calls = calls_per_step[kern]
        t_launch = per_step_ms[kern] * 1e-3 / calls
        bx3_now = _lib_arith() == 1 and kern in BX3_OPS
        if bound == "mfma":
            # bf16 x 3 entry points: fp32-EQUIVALENT flops against the fp32-equivalent matrix peak of that arithmetic
            # (six bf16 MFMA products per fp32 product: dense bf16 peak / 6)
            ach, peak, unit = fl / calls / t_launch / 1e12, (BF16_MFMA_PEAK_TF / 6.0 if bx3_now else F32_MFMA_PEAK_TF), "TFLOP/s"
        else:
            ach, peak, unit = by / calls / t_launch / 1e9, HBM_PEAK_GBS, "GB/s"
        traffic = rocprof_us = rocprof_ms_step = frac_kernel = None
        main = MAIN_KERNEL.get(kern, kern)

        def base(name):  # "void r3d_x_kernel<5>(float const*, ..." -> "r3d_x_kernel"
            return name.split("(")[0].split("<")[0].replace("void ", "").strip()
        # the two kNN entry points run instances of ONE kernel template: told apart by their template arguments
        by_template = {"knn_topk": ("r3d_knn_append_kernel<4, 128, 1,",), "knn_topk_l2": ("r3d_knn_append_kernel<8, 384, 2,",)}

        def mine(name):
            if kern in by_template:
                return name.replace("void ", "").startswith(by_template[kern])
            return bool(base(name)) and base(name) in main
        if pmc is not None:
            hit = [v for k, v in pmc.items() if mine(k)]
            if hit:  # the SUM over the named kernels (one launch of each per call of the entry point)
                traffic = 1024.0 * sum(FETCH_SIZE_WIDE_READ_FACTOR * v["fetch_kb_per_launch"] + v["write_kb_per_launch"] for v in hit)
        if prof is not None:
            hits = [v for k, v in prof.items() if k != "_steps" and mine(k)]
            if hits:
                rocprof_us = max(v[2] for v in hits)  # average duration of the entry point's longest kernel
                if prof.get("_steps"):
                    # the entry point's main kernels in the committed kernel trace of the same workload and schedule: their
                    # time per step, and the roofline fraction from kernel time alone (no helper kernels, no launch gaps)
                    rocprof_ms_step = sum(v[1] for v in hits) / prof["_steps"]
                    frac_kernel = (fl / (rocprof_ms_step * 1e-3) / 1e12 / peak) if bound == "mfma" else \
                        (by / (rocprof_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS)
        useful = None
        if kern + "_useful" in ("edgeconv_bwd_useful",):  # flops the math needs (recompute excluded) over the same time
            fu = algorithmic_work(kern + "_useful", cfg, n_nodes, cg_mean, train)[0] * E
            useful = fu / calls / t_launch / 1e12 / peak
        roof = dict(kernel="%s (entry point %s, %d episodes per launch)" % (main, kern, E), bound=bound, achieved=ach, peak=peak,
                    unit=unit, frac=ach / peak, useful_frac=useful if useful is not None else ach / peak,
                    peak_note=("fp32-equivalent peak of the bf16 x 3 arithmetic: 2500 TFLOP/s dense bf16 / 6 products per fp32 "
                               "product" if (bound == "mfma" and bx3_now) else None), traffic=traffic, avg_launch_ms=t_launch * 1e3, launches_per_step=calls,
                    algorithmic_gflop_per_launch=fl / calls / 1e9, algorithmic_mb_per_launch=by / calls / 1e6,
                    hbm_gbs=by / calls / t_launch / 1e9, fp32_tflops=fl / calls / t_launch / 1e12,
                    rocprofv3_kernel_us=rocprof_us, rocprofv3_main_kernels_ms_per_step=rocprof_ms_step,
                    event_ms_per_step=per_step_ms[kern], frac_kernel_time_only=frac_kernel,
                    rocprofv3_summary=prof_name, pmc_summary=pmc_name,
                    note="one HIP event pair per entry-point call on the launch stream (launches of the batched step do not "
                         "overlap; the pair includes the entry point's small helper kernels); traffic = %.0f x FETCH_SIZE + "
                         "WRITE_SIZE of the main kernel from the committed PMC passes" % FETCH_SIZE_WIDE_READ_FACTOR)
    # every entry point against both ceilings (north_star: HBM GB/s for kNN / EdgeConv, MFMA utilisation for attention)
    from r3dfsseg_amd import _lib as _l0
    bx3_attention = _l0.load().r3d_get_matrix_arith() == 1
    rooflines = {"_note": "HIP event pairs around every entry-point call of a live step (%d episodes per launch): hbm_gbs is "
                          "algorithmic bytes / device time, not HBM traffic" % E}
    for op, ms in per_step_ms.items():
        f_, b_, bnd, _ = algorithmic_work(op, cfg, n_nodes, cg_mean, train)
        f_, b_ = f_ * E, b_ * E
        sec = ms * 1e-3
        rooflines[op] = dict(bound=bnd, ms_per_step=round(ms, 4), calls_per_step=calls_per_step[op],
                             hbm_gbs=round(b_ / sec / 1e9, 1), frac_hbm=round(b_ / sec / 1e9 / HBM_PEAK_GBS, 4),
                             fp32_tflops=round(f_ / sec / 1e12, 2), frac_mfma=round(f_ / sec / 1e12 / F32_MFMA_PEAK_TF, 4))
        if op in BX3_OPS and bx3_attention:
            # six bf16 MFMA products per fp32 product: the matrix-core work actually issued, against the dense bf16 peak
            rooflines[op].update(matrix_arith="bf16 x 3", bf16_tflops_issued=round(6 * f_ / sec / 1e12, 1),
                                 frac_mfma=round(6 * f_ / sec / 1e12 / BF16_MFMA_PEAK_TF, 4),
                                 frac_mfma_note="issued bf16 MFMA flops (6 per fp32 product) / 2500 TFLOP/s dense bf16 peak; "
                                                "fp32_tflops is the useful fp32-equivalent rate")
    breakdown = {k: round(v, 4) for k, v in sorted(per_step_ms.items(), key=lambda kv: -kv[1])}
    # the whole step: useful fp32-equivalent flops of every matrix-shaped entry point (recomputes excluded) over the step time
    step_flops = 0.0
    for op in per_step_ms:
        f_u = algorithmic_work(op + "_useful" if op == "edgeconv_bwd" else op, cfg, n_nodes, cg_mean, train)
        if f_u[2] == "mfma":
            step_flops += f_u[0] * E

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(S, cfg, args.workload, extra, train, batched)

    eps = args.steps * E * world / elapsed
    from r3dfsseg_amd import _lib as _l
    matrix_arith = ("self-attention, 1x1-convolution GEMMs (forward and input gradient), weight-gradient GEMMs and the three edge "
                    "GEMMs of the EdgeConv backward: fp32 operands as three bf16 pieces on the bf16 MFMA, six products per block, "
                    "fp32 accumulate (error against float64 as the fp32 kernels: tools/gemm_accuracy.py).  kNN on 64 channels: a "
                    "bf16 bound pass FILTERS the candidates, every emitted score is the exact fp32 fmaf chain (bit-identical "
                    "to the all-pairs fp32 pass).  Index-deciding kernels (kNN scores, the EdgeConv forward edge GEMM "
                    "whose max-pool winners route the gradient, FPS) and the 9-channel input layer: fp32"
                    if _l.load().r3d_get_matrix_arith() == 1 else "fp32 MFMA everywhere")
    dtype_str = ("f32 (index-free GEMMs -- self-attention, 1x1 convolutions, weight gradients, the EdgeConv backward's edge GEMMs "
                 "-- as bf16x3 split products on the bf16 MFMA with fp32 accumulate; every index-deciding result: fp32)"
                 if _l.load().r3d_get_matrix_arith() == 1 else "f32")
    out = {
        "metric": "episodes/sec %s %d-way %d-shot %d-pt (MPTI+attention, %s)" % (
            "ScanNet" if args.workload == "C" else "S3DIS", cfg["n_way"], cfg["k_shot"], N,
            "train step: forward+backward+grad all-reduce+Adam" if train else "eval forward"),
        "value": eps, "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype_str, "matrix_arith": matrix_arith, "data": "synthetic",
        "config": {"workload": "%s: %d-way %d-shot %d pts, %d clouds/episode, %d episode(s)/step/rank (%s), mode=%s" % (
            args.workload, cfg["n_way"], cfg["k_shot"], N, B, E,
            "%d per launch sequence (episode-batched kernels)" % Bsz if batched else "eager launches, one at a time", args.mode),
            "episodes_per_step": E * world, "episodes_per_rank": E, "episodes_per_launch_sequence": Bsz,
            "optimiser_steps_before_timing": (steps_done - args.steps) if train else 0,
            "launch": ("one captured hipGraph per step (batched.BatchGraph: the batch's ~450 launches frozen; same kernels, "
                       "bit-identical results)" if (train and batched and trainer.batch_graph) else "eager launches")},
        "valid": int(n_invalid.item()) == 0, "invalid": invalid,
        "roofline": roof, "roofline_cg_iteration": cg_roof, "rooflines": rooflines, "cpu_baseline": cpu,
        "entry_point_ms_per_step": breakdown,
        "step_fp32_equiv_tflops": step_flops / (elapsed / args.steps) / 1e12,
        "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
        "lp_cg_iterations": {"mean": cg_mean, "max": cg_max},
        "train_steps_redone": {"count": getattr(trainer, "n_redone", 0) if trainer is not None else 0, "notes": redone_notes},
    }
    out.update(extra)
    if not out["valid"]:
        print("bench.py: WARNING, the line below is flagged invalid: %s" % "; ".join(invalid), file=sys.stderr)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(S, cfg, workload, extra, train, batched):
    """The CPU oracle (kind "port": C + torch-CPU restatement of the reference path, oracle/) on this box's host cores,
    on a bounded sample of the same workload: eval forward (1 warm-up, 5 timed) and one-episode training steps (forward,
    backward through the dense closed-form label propagation, Adam; 1 warm-up, 3 timed)."""
    # the GPU box hands one GPU a 16-core CPU share (os.cpu_count() reports the whole host)
    ncores = min(os.cpu_count() or 1, int(os.environ.get("R3D_CPU_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    from oracle import r3d_oracle as O
    torch.set_num_threads(ncores)
    sd = S.make_state_dict(cfg, 123)
    data, _ = S.make_episode(cfg, seed=0, noise_ratio=0.2, train=True)
    sx, sy, qx, qy = data[:4]
    O.knn(qx[:1, :, :256], 4)  # load + warm the C library
    times = []
    for i in range(6):
        c0 = time.perf_counter()
        with torch.no_grad():
            O.mpti_forward(sd, cfg, sx, sy, qx, qy)
        if i:
            times.append(time.perf_counter() - c0)
    times.sort()
    ev_min, ev_med = times[0], times[len(times) // 2]
    # training step of the reference's schedule (mpti_learner.py:60-72): one episode, loss = lp + 0.1 contrast, Adam
    params = {k: v.clone().requires_grad_() for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sdt = dict(sd)
    sdt.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    ttimes = []
    for i in range(4):
        c0 = time.perf_counter()
        out = O.mpti_forward(sdt, cfg, sx, sy, qx, qy, gt_support_y=data[6], gt_query_y=data[7], train=True,
                             support_flag=data[10], new_stats={})
        loss = out[1] + 0.1 * out[2]
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i:
            ttimes.append(time.perf_counter() - c0)
    ttimes.sort()
    tr_med = ttimes[len(ttimes) // 2]
    gpu_eval = extra.get("eval_forward_episodes_per_sec")
    gpu_single = extra.get("single_episode_eager_step_episodes_per_sec")
    return dict(value=1.0 / ev_med, unit="episodes/s", cores=ncores, kind="port",
                sample="%s episodes on the CPU oracle (C + torch-CPU, %d threads): EVAL FORWARD 1 warm-up + 5 timed (value = 1 / median), "
                       "TRAIN STEP (forward + backward + Adam, dropout off) 1 warm-up + 3 timed" % (workload, ncores),
                eval_forward={"episodes_per_sec_median": 1.0 / ev_med, "episodes_per_sec_best": 1.0 / ev_min, "timed": 5, "warmup": 1},
                train_step={"episodes_per_sec_median": 1.0 / tr_med, "timed": 3, "warmup": 1},
                compare_with={"eval_forward": "eval_forward_episodes_per_sec", "train_step": "single_episode_eager_step_episodes_per_sec (same schedule) / value"},
                gpu_eval_forward_over_cpu=(gpu_eval * ev_med) if gpu_eval else None,
                gpu_single_episode_train_step_over_cpu=(gpu_single * tr_med) if gpu_single else None)


if __name__ == "__main__":
    main()
