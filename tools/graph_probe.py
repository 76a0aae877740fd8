"""Does whole-episode hipGraph capture work through the ctypes launches, and do 2-4 concurrent replays overlap?"""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import ops, synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
dev = torch.device("cuda", 0)
cfg = S.workload_cfg("S")
G = int(os.environ.get("G", "4"))
models, inputs, graphs, outs, streams = [], [], [], [], []
for s in range(G):
    m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.to(dev).eval()
    m._lp_budget = 40; m._lp_post = lambda hb: None
    data, _ = S.make_episode(cfg, seed=s, noise_ratio=0.2, train=True)
    models.append(m); inputs.append([t.to(dev) for t in data[:4]]); streams.append(torch.cuda.Stream())
with torch.no_grad():
    for s in range(G):
        for _ in range(3): models[s](*inputs[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): models[0](*inputs[0])
    torch.cuda.synchronize()
    print("eager          %.3f ms/episode" % ((time.perf_counter() - t0) / 20 * 1e3))
    for s in range(G):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o = models[s](*inputs[s])
        graphs.append(g); outs.append(o)
    torch.cuda.synchronize()
    ref = models[0](*inputs[0])[0].clone()
    graphs[0].replay(); torch.cuda.synchronize()
    print("graph == eager:", torch.equal(ref, outs[0][0]))
    for n in (1, 2, G):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            for s in range(n):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("graphs x%d       %.3f ms/episode (host issue %.3f)" % (n, (t2 - t0) / 20 / n * 1e3, (t1 - t0) / 20 / n * 1e3))
