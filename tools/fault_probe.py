"""Replay patterns of the episode hipGraphs, each stage in its own process with progress markers after syncs:
run, synchronise, (eager work / weight update), run again.  With memset / memcpy nodes inside the captured
sequence the second run faulted on ROCm 7.2 (profiles/r01_experiments.md); with kernel nodes only every stage
passes."""
import os, sys, subprocess
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def stage(name):
    import torch
    from r3dfsseg_amd import synthetic as S
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    def mark(s):
        torch.cuda.synchronize(); print(name, s, flush=True)
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    train = name.startswith("T")
    m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train(train)
    m.att_learner.dropout.p = 0.0
    eps = []
    for e in range(3):
        data, _ = S.make_episode(cfg, seed=40 + e, noise_ratio=0.2, train=True)
        eps.append([t.cuda() for t in data])
    n = 11 if train else 4
    g = EpisodeGraphs(m, eps[0][:n], n_slots=int(os.environ.get("SLOTS", "2")), train=train, lp_budget=150)
    mark("captured")
    g.run([ep[:n] for ep in eps]); mark("run1")
    if name == "E1":
        with torch.no_grad(): m(*eps[1][:4])
        mark("eager")
    elif name == "E2":
        with torch.no_grad(): m(*eps[1][:4], lp_iters=m.lp_max_iter)
        mark("eager_lp_iters")
    elif name == "E3":
        with torch.no_grad():
            for p in m.parameters(): p.mul_(1.03)
        mark("weights scaled")
    elif name == "T1":
        pass
    elif name == "T2":
        x = torch.randn(1 << 20, device="cuda"); y = (x * 2).sum().item(); del x
        mark("eager allocs")
    g.run([ep[:n] for ep in eps]); mark("run2")
    print(name, "check", g.check(), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        stage(sys.argv[1])
    else:
        for n, env in [("T1", dict(SLOTS="1")), ("E1", dict(SLOTS="1")), ("T1", dict()), ("T2", dict()), ("E1", dict()),
                       ("E2", dict()), ("E3", dict())]:
            r = subprocess.run([sys.executable, __file__, n], capture_output=True, text=True, timeout=200, env=dict(os.environ, **env))
            print("==", n, env, "rc", r.returncode, "|", " / ".join(l for l in r.stdout.splitlines()), "|",
                  " ".join(l for l in r.stderr.splitlines() if "fault" in l.lower())[:200], flush=True)
