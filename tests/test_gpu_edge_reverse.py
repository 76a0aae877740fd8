"""Reverse neighbour list (csrc/edgeconv_train.hip: r3d_edge_reverse) against a stable sort on the host, and the
run-to-run bit-identity of the EdgeConv input gradient it makes possible (the first version scattered with float
atomics).  Reference semantics: autograd's scatter-add through torch.gather in models/dgcnn.py:38."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reverse(idx):
    from r3dfsseg_amd import _lib, ops
    lib = _lib.load()
    B, N, K = idx.shape
    ws = torch.full((lib.r3d_edge_reverse_ws_words(B, N, K),), -7, device="cuda", dtype=torch.int32)
    _lib.check(lib.r3d_edge_reverse(ops._p(idx), B, N, K, ops._p(ws), ws.numel(), ops._st()))
    torch.cuda.synchronize()
    ws = ws.cpu().numpy()
    return ws[:B * N + 1], ws[B * N + 1:B * N + 1 + B * N * K]


def _want(idx):
    B, N, K = idx.shape
    tgt = (np.clip(idx, 0, N - 1) + np.arange(B)[:, None, None] * N).reshape(-1)
    order = np.argsort(tgt, kind="stable")                      # ascending edge id inside a target
    ptr = np.searchsorted(tgt[order], np.arange(B * N + 1), side="left")
    return ptr.astype(np.int32), order.astype(np.int32)


@pytest.mark.parametrize("B,N,K,kind", [(3, 72, 20, "knn"), (12, 2048, 20, "knn"), (2, 4096, 20, "knn"), (2, 1000, 8, "hubs"),
                                        (1, 2048, 32, "one_target"), (2, 640, 4, "garbage")])
def test_reverse_list_equals_stable_sort(B, N, K, kind):
    rs = np.random.RandomState(B * 1000 + N + K)
    if kind == "knn":        # distinct neighbours per point, self included: what r3d_knn_topk writes
        idx = np.stack([np.stack([np.r_[i, rs.permutation(N)[:K - 1]] for i in range(N)]) for _ in range(B)])
    elif kind == "hubs":     # a few points named by most lists: segments far longer than a wave
        idx = rs.randint(0, 12, (B, N, K)) * (N // 12)
    elif kind == "one_target":  # every edge names point 5: more entries than the LDS buffer -> ordered compaction
        idx = np.full((B, N, K), 5)
    else:                    # ids outside the cloud are clamped, as the kernels do everywhere
        idx = rs.randint(-50, N + 50, (B, N, K))
    idx_t = torch.from_numpy(idx.astype(np.int32)).cuda().contiguous()
    ptr, rev = _reverse(idx_t)
    wptr, wrev = _want(idx)
    assert np.array_equal(ptr, wptr)
    assert np.array_equal(rev, wrev)


def test_edgeconv_input_gradient_is_bit_identical_run_to_run():
    from types import SimpleNamespace
    from r3dfsseg_amd import train_ops as T
    rs = np.random.RandomState(4)
    B, N, C, K = 4, 512, 64, 20
    x = torch.from_numpy(rs.randn(B * N, C).astype(np.float32)).cuda()
    idx = torch.from_numpy(np.stack([[rs.permutation(N)[:K] for _ in range(N)] for _ in range(B)]).astype(np.int32)).cuda()
    conv1, conv2 = torch.nn.Conv2d(2 * C, 64, 1, bias=False).cuda(), torch.nn.Conv2d(64, 64, 1, bias=False).cuda()
    bn1, bn2 = torch.nn.BatchNorm2d(64).cuda().train(), torch.nn.BatchNorm2d(64).cuda().train()
    ec = SimpleNamespace(layer=[conv1, bn1, None, conv2, bn2])
    R = torch.from_numpy(rs.randn(B * N, 64).astype(np.float32)).cuda()
    outs = []
    for _ in range(3):
        out = torch.empty(B * N, 64, device="cuda")
        saved = T.edgeconv_train_fwd(x, idx, ec, B, N, out)
        dx = torch.zeros(B * N, C, device="cuda")
        grads = T.edgeconv_train_bwd(saved, R, B, N, dx)
        torch.cuda.synchronize()
        outs.append([dx.clone()] + [g.clone() for g in grads])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_short_workspace_is_an_error_not_an_overrun():
    from r3dfsseg_amd import _lib, ops
    lib = _lib.load()
    idx = torch.zeros(2, 64, 8, device="cuda", dtype=torch.int32)
    ws = torch.empty(lib.r3d_edge_reverse_ws_words(2, 64, 8) - 1, device="cuda", dtype=torch.int32)
    assert lib.r3d_edge_reverse(ops._p(idx), 2, 64, 8, ops._p(ws), ws.numel(), ops._st()) != 0
    assert b"workspace" in lib.r3d_last_error_string()
    hb = ops.HeadBuffers(2, 1, 256, 512, 100, 200, 192, "cuda")
    hb.desc[ops.HD_N_PROTO] = 10
    hb.desc[ops.HD_N_NODES] = 522
    nbr = torch.zeros(hb.n_cap, hb.kp1, device="cuda", dtype=torch.int32)
    short = hb.lp_ws[:hb.lp_ws.numel() - 64]
    rc = lib.r3d_label_propagate(ops._p(hb.nodes), hb.nodes.stride(0), hb.D, ops._p(nbr), hb.kp1, ops._p(hb.Y),
                                 ops._p(hb.desc[ops.HD_N_NODES:]), ops._p(hb.desc[ops.HD_N_PROTO:]), hb.n_cap, 1.0, 0.99, 10, 1e-6,
                                 ops._p(hb.Z), ops._p(short), short.numel(), ops._p(hb.stats), ops._st())
    assert rc != 0 and b"workspace" in lib.r3d_last_error_string()
