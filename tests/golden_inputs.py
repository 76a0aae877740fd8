"""Inputs of the golden vectors (same seeds as oracle/gen_golden.py)."""
import numpy as np
import torch

from r3dfsseg_amd import synthetic as S


def golden_inputs():
    r = np.random.RandomState
    return dict(
        x9=torch.from_numpy(r(11).randn(2, 9, 512).astype(np.float32)),
        x64=torch.from_numpy(r(12).randn(2, 64, 512).astype(np.float32)),
        pc=torch.from_numpy(np.stack([S._cloud(r(13 + i), 512, 0.0).T for i in range(2)]).copy()),
        x256=torch.from_numpy((r(14).randn(2, 256, 512) * 0.5).astype(np.float32)),
    )
