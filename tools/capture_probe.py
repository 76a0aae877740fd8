"""Which part of the training step breaks hipGraph capture?  Each stage runs in its own process."""
import os, sys, subprocess
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def stage(name):
    import torch
    from r3dfsseg_amd import synthetic as S, train_ops as T, ops
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    dev = "cuda"
    if name.startswith("graphs"):
        from r3dfsseg_amd.episode_graph import EpisodeGraphs
        cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
        m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
        data, _ = S.make_episode(cfg, seed=3, noise_ratio=0.2, train=True)
        ep = [t.cuda() for t in data]
        if "pre" in name:
            out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10])
            (out[1] + 0.1 * out[2]).backward()
        g = EpisodeGraphs(m, ep, n_slots=int(name[-1]), train=True, lp_budget=150)
        g.run([ep, ep, ep]); torch.cuda.synchronize()
        print(name, "OK", g.check(), flush=True)
        return
    if name == "torch_linear":
        lin = torch.nn.Linear(64, 64).cuda(); x = torch.randn(32, 64, device=dev)
        def run():
            lin(x).sum().backward()
    else:
        cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
        m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().train()
        m._slot.fixed_budget = 100
        m._slot.seed_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        data, _ = S.make_episode(cfg, seed=3, noise_ratio=0.2, train=True)
        ep = [t.cuda() for t in data]
        def fwd():
            return m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10])
        if name == "fwd":
            def run():
                with torch.no_grad(): fwd()
        elif name == "fwd_bwd":
            def run():
                out = fwd(); (out[1] + 0.1 * out[2]).backward()
        elif name == "enc_bwd":
            def run():
                f = T.get_features_train(m, ep[2], 0); f.sum().backward()
        elif name == "lp_bwd":
            def run():
                out = fwd(); out[1].backward()
        elif name == "contrast_bwd":
            def run():
                out = fwd(); out[2].backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): run()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    print(name, "OK", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        stage(sys.argv[1])
    else:
        for n in ["graphs_pre2"]:
            r = subprocess.run([sys.executable, __file__, n], capture_output=True, text=True, timeout=200, env=dict(os.environ, PYTHONFAULTHANDLER='1'))
            print(n, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], "|", "\n".join(l for l in r.stderr.splitlines() if "episode_graph" in l or "Fatal" in l or "Error" in l)[:1500], flush=True)
