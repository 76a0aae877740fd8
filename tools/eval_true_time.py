"""Eager forward time with and without clean-shot detection (eval=True, mpti.py:440-442)."""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import synthetic as S
from r3dfsseg_amd.mpti import MPTI_SelfAtten
cfg = S.workload_cfg("S")
m = MPTI_SelfAtten(SimpleNamespace(**cfg)); m.load_state_dict(S.make_state_dict(cfg, 123)); m.cuda().eval()
data, _ = S.make_episode(cfg, seed=5, noise_ratio=0.4)
ep = [t.cuda() for t in data[:4]]
for flag in (False, True):
    with torch.no_grad():
        for _ in range(5): m(*ep, eval=flag)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): m(*ep, eval=flag)
        torch.cuda.synchronize()
    print("eval=%s: %.2f ms per episode" % (flag, (time.perf_counter() - t0) / 30 * 1e3))
