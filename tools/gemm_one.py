"""One point-wise GEMM shape launched 20 times (for PMC passes).  usage: gemm_one.py M K Co"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from r3dfsseg_amd import ops
M, K, Co = (int(a) for a in sys.argv[1:4])
X = torch.randn(M, K, device="cuda"); W = torch.randn(Co, K, device="cuda") * 0.1
for _ in range(20): ops.pointwise_conv(X, W)
torch.cuda.synchronize()
