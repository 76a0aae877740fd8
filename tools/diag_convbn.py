import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r3dfsseg_amd import _lib, ops, train_ops as T
lib = _lib.load()
rs = np.random.RandomState(0)
for (M, K, C) in [(24576, 192, 512), (24576, 512, 256), (24576, 64, 128)]:
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32)); x = torch.relu(x) * 1.3 + 0.05 * x
    W = torch.from_numpy((rs.randn(C, K) / np.sqrt(K)).astype(np.float32))
    g = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)); b = torch.from_numpy(rs.uniform(-0.2, 0.2, C).astype(np.float32))
    R = torch.from_numpy(rs.randn(M, C).astype(np.float32))
    xr, Wr = x.double().requires_grad_(), W.double().requires_grad_()
    bnr = torch.nn.BatchNorm1d(C).double(); bnr.weight.data = g.double(); bnr.bias.data = b.double()
    y = torch.nn.functional.leaky_relu(bnr(xr @ Wr.t()), 0.2); (y * R.double()).sum().backward()
    rel = lambda a, r: ((a.double().cpu() - r).abs().max() / r.abs().max()).item()
    for name, arith in (("fp32", 0), ("bx3", 1)):
        lib.r3d_set_matrix_arith(arith)
        bn = torch.nn.BatchNorm1d(C); bn.weight.data = g.clone(); bn.bias.data = b.clone(); bn = bn.cuda()
        yg, saved = T.conv_bn_fwd(x.cuda(), W.cuda(), bn, ops.ACT_LRELU)
        dW, dg, db, dbias, dX = T.conv_bn_bwd(saved, R.cuda())
        flips = ((yg.cpu().double() > 0) != (y.detach() > 0)).sum().item()
        print("%6d x %3d -> %3d %-5s y %.2e  dW %.2e  dX %.2e  dgamma %.2e  dbeta %.2e  sign flips %d   mean %.3e var-ish" % (
            M, K, C, name, rel(yg, y.detach()), rel(dW, Wr.grad), rel(dX, xr.grad), rel(dg, bnr.weight.grad), rel(db, bnr.bias.grad), flips,
            rel(bn.running_mean, bnr.running_mean)))
