"""Thin torch-tensor wrappers over the C ABI (include/r3d.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and
tensor views.  All compute happens in libr3d_hip.so; there is no eager fallback.
"""
import ctypes

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
SCORE_DGCNN, SCORE_L2 = 0, 1
HD_SEG_COUNT, HD_SEG_M, HD_SEG_POFF, HD_N_PROTO, HD_N_NODES, HD_FPS_TIMEOUT = 0, 8, 16, 24, 25, 26
HEAD_FPS_ONE_LAUNCH = 1


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class KernelTimer:
    """HIP-event timing of selected entry points on the stream they are launched on
    (bench.py's roofline leg).  Disabled (None) by default: zero overhead.

    repeat = 0: one event pair around every call (includes the event packets and any host launch gap).
    repeat = R > 0: after a timed region ran, its library calls are launched again R times back to back
    between ONE event pair and the region is priced at elapsed / R: device time of the launches, with the
    event and host overhead amortised (the calls are idempotent: same inputs, same outputs)."""

    def __init__(self, names, repeat=0):
        self.names = set(names)
        self.events = {n: [] for n in names}
        self.repeat = repeat
        if repeat:
            _lib.record_calls(True)

    def close(self):
        if self.repeat:
            _lib.record_calls(False)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, ev in self.events.items():
            ms = [a.elapsed_time(b) / r for a, b, r in ev]
            out[n] = dict(launches=len(ms), total_ms=float(sum(ms)), avg_ms=float(sum(ms) / max(len(ms), 1)))
        return out


_TIMER = None


def set_timer(timer):
    global _TIMER
    _TIMER = timer


class _timed:
    def __init__(self, name):
        self.on = _TIMER is not None and name in _TIMER.names
        self.name = name

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            if _TIMER.repeat:
                _lib._call_log = []
            else:
                self.a.record()

    def __exit__(self, *exc):
        if self.on:
            if _TIMER.repeat:
                calls, _lib._call_log = _lib._call_log, None
                if exc[0] is None and calls:
                    self.a.record()
                    for _ in range(_TIMER.repeat):
                        for fn, args in calls:
                            fn(*args)
                    self.b.record()
                    _TIMER.events[self.name].append((self.a, self.b, _TIMER.repeat))
            else:
                self.b.record()
                _TIMER.events[self.name].append((self.a, self.b, 1))


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t):
    """(rows, ld) of a 2-D fp32 row-major view (unit column stride)."""
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, \
        "expected a 2-D fp32 CUDA tensor with unit column stride"
    return t.shape[0], t.stride(0)


def cm_to_pm(x, out=None):
    """(B, C, N) channel-major -> (B*N, C) point-major."""
    B, C, N = x.shape
    x = x.contiguous().float()
    if out is None:
        out = torch.empty(B * N, C, device=x.device, dtype=torch.float32)
    _, ld = _rows(out)
    _lib.check(_lib.load().r3d_cm_to_pm(_p(x), B, C, N, _p(out), ld, _st()))
    return out


def pm_to_cm(x_pm, B, N):
    """(B*N, C) point-major view -> (B, C, N) contiguous."""
    M, ld = _rows(x_pm)
    C = x_pm.shape[1]
    assert M == B * N
    out = torch.empty(B, C, N, device=x_pm.device, dtype=torch.float32)
    _lib.check(_lib.load().r3d_pm_to_cm(_p(x_pm), ld, B, C, N, _p(out), _st()))
    return out


def copy_cols(src, dst):
    M, lds = _rows(src)
    M2, ldd = _rows(dst)
    assert M == M2 and src.shape[1] == dst.shape[1]
    _lib.check(_lib.load().r3d_copy_cols(_p(src), lds, _p(dst), ldd, M, src.shape[1], _st()))
    return dst


def knn(x_pm, B, N, k, mode=SCORE_DGCNN, n_valid=None, return_scores=False, x_cm=None, status=None):
    """x_pm (B*N, C) -> idx (B, N, k) int32, best first.  x_cm: optional (B, C, N) channel-major
    copy of the same points (saves the internal transpose of the streamed k <= 32 kernel)."""
    M, ld = _rows(x_pm)
    C = x_pm.shape[1]
    assert M == B * N
    dev = x_pm.device
    norm = torch.empty(_lib.load().r3d_knn_norm_ws_words(B, N), device=dev, dtype=torch.float32)
    idx = torch.empty(B, N, k, device=dev, dtype=torch.int32)
    sc = torch.empty(B, N, k, device=dev, dtype=torch.float32) if return_scores else None
    cm_ws = None
    if k <= 32 and C <= 64:
        if x_cm is None:
            cm_ws = torch.empty(B * C * _lib.load().r3d_cm_pitch(N), device=dev, dtype=torch.float32)
        else:
            assert x_cm.is_contiguous() and x_cm.shape == (B, C, N) and x_cm.dtype == torch.float32
    elif status is not None:  # large-k streamed kernel (status bit 0 = overflow -> redo with status=None)
        assert status.dtype == torch.int32
        if x_cm is None:
            cm_ws = torch.empty(B * C * _lib.load().r3d_cm_pitch(N), device=dev, dtype=torch.float32)
    else:
        x_cm = None
    lib = _lib.load()
    split_ws, split_words = None, 0
    if k > 32 and status is not None and 2 * B * ((N + 31) // 32) <= 256:  # the large-k kernel may split the candidate axis
        split_words = lib.r3d_knn_split_ws_words(B, N, k)
        split_ws = torch.empty(split_words, device=dev, dtype=torch.float32)
    with _timed("knn_topk_l2" if mode == SCORE_L2 else "knn_topk"):
        _lib.check(lib.r3d_knn_topk_split(_p(x_pm), ld, _p(x_cm), B, N, C, k, mode, _p(n_valid), _p(norm), _p(cm_ws),
                                          _p(idx), _p(sc), _p(status), _p(split_ws), split_words, _st()))
    return (idx, sc) if return_scores else idx


def pointwise_conv(x_pm, W, scale=None, shift=None, act=ACT_NONE, out=None):
    """act(scale * (x W^T) + shift); W (Co, K) contiguous."""
    M, ldx = _rows(x_pm)
    K = x_pm.shape[1]
    Co = W.shape[0]
    assert W.is_contiguous() and W.shape[1] == K and W.dtype == torch.float32
    if out is None:
        out = torch.empty(M, Co, device=x_pm.device, dtype=torch.float32)
    M2, ldo = _rows(out)
    assert M2 == M and out.shape[1] == Co
    with _timed("pointwise_conv"):
        _lib.check(_lib.load().r3d_pointwise_conv(_p(x_pm), ldx, _p(W), M, K, Co, _p(scale), _p(shift), act,
                                                  _p(out), ldo, _st()))
    return out


def edgeconv(PQ, idx, W2, s2, t2, out, B, N, want_argmax=False):
    K = idx.shape[-1]
    assert PQ.is_contiguous() and PQ.shape == (B * N, 128) and idx.is_contiguous() and idx.dtype == torch.int32
    assert W2.is_contiguous() and W2.shape == (64, 64)
    M, ldo = _rows(out)
    assert M == B * N and out.shape[1] == 64
    am = torch.empty(B * N, 64, device=PQ.device, dtype=torch.int32) if want_argmax else None
    with _timed("edgeconv"):
        _lib.check(_lib.load().r3d_edgeconv_fwd(_p(PQ), _p(idx), _p(W2), _p(s2), _p(t2), _p(out), ldo, B, N, K,
                                                _p(am), _st()))
    return am


def attention(qkv, B, N, out, want_lse=False):
    M, ld = _rows(qkv)
    M2, ldo = _rows(out)
    assert M == B * N and M2 == M and qkv.shape[1] == 192 and out.shape[1] == 64
    lse = torch.empty(M, device=qkv.device, dtype=torch.float32) if want_lse else None
    lib = _lib.load()
    ws = torch.empty(lib.r3d_attention_ws_words(B, N), device=qkv.device, dtype=torch.float32)
    with _timed("attention"):
        _lib.check(lib.r3d_attention_fwd(_p(qkv), ld, B, N, _p(out), ldo, _p(lse), _p(ws), _st()))
    return lse


class HeadBuffers:
    """Device buffers of one episode's transductive head (capacity sized, no host sync)."""

    def __init__(self, n_way, k_shot, N, n_q_pts, k_sub, k_connect, D, device):
        lib = _lib.load()
        self.n_way, self.k_shot, self.N, self.n_q_pts, self.k_sub, self.kp1, self.D = \
            n_way, k_shot, N, n_q_pts, k_sub, k_connect + 1, D
        self.n_cap = (n_way + 1) * k_sub + n_q_pts
        assert lib.r3d_head_desc_words() == 32
        i32 = dict(device=device, dtype=torch.int32)
        f32 = dict(device=device, dtype=torch.float32)
        self.desc = torch.zeros(32, **i32)
        self.nodes = torch.empty(self.n_cap, D, **f32)
        self.Y = torch.empty(self.n_cap, 4, **f32)
        self.Z = torch.empty(self.n_cap, 4, **f32)
        self.proto_ws = torch.empty(lib.r3d_head_proto_ws_words(n_way, k_shot, N), **i32)
        self.lp_ws = torch.empty(lib.r3d_lp_ws_words(self.n_cap, self.kp1), **i32)
        self.assign = torch.empty(2 * n_way * k_shot * N, **i32)
        self.cluster_count = torch.zeros(self.n_cap, **i32)
        self.stats = torch.zeros(2, **i32)
        self.knn_status = torch.zeros(1, **i32)
        self.stats_bwd = torch.zeros(2, **i32)
        # all FPS rounds in one persistent launch: safe while (episodes in flight) x fps_blocks workgroups stay
        # co-resident (~384 of the chip's 512 slots for this kernel); episode_graph.EpisodeGraphs decides per slot
        self.fps_one_launch = True
        self.fps_blocks = (n_way * k_shot * N + 255) // 256 + n_way + 1
        off = (ctypes.c_long * 6)()
        lib.r3d_head_proto_ws_offsets(n_way, k_shot, N, off)
        self.ws_off = list(off)
        lib.r3d_lp_ws_offsets(self.n_cap, self.kp1, off)
        self.lp_off = dict(zip(("row_ptr", "col", "val", "dinv", "agg", "cg"), off))

    def csr(self):
        """(n, row_ptr (n+1) int64, col (nnz) int64, val (nnz) fp32) of the normalised graph S the last
        r3d_label_propagate left in the workspace (synchronises; tests, tools and bench.py's byte counts)."""
        n = int(self.desc[HD_N_NODES].item())
        o = self.lp_off
        row_ptr = self.lp_ws[o["row_ptr"]:o["row_ptr"] + n + 1].to(torch.int64)
        nnz = int(row_ptr[-1].item())
        col = self.lp_ws[o["col"]:o["col"] + (nnz + 1) // 2].view(torch.int16)[:nnz].to(torch.int64) & 0xffff
        val = self.lp_ws[o["val"]:o["val"] + nnz].view(torch.float32)
        return n, row_ptr, col, val


def head_prototypes(hb, support_y, shot_keep, sfeat_pm, sfeatT, qfeat_pm):
    S = hb.n_way * hb.k_shot
    M, ldf = _rows(sfeat_pm)
    Mq, ldq = _rows(qfeat_pm)
    assert M == S * hb.N and Mq == hb.n_q_pts and sfeatT.is_contiguous() and sfeatT.shape == (S, hb.D, hb.N)
    assert support_y.dtype == torch.int32 and support_y.is_contiguous() and support_y.numel() == S * hb.N
    with _timed("head_prototypes"):
        _lib.check(_lib.load().r3d_head_prototypes(
            _p(support_y), _p(shot_keep), _p(sfeat_pm), ldf, _p(sfeatT), _p(qfeat_pm), ldq, hb.n_way, hb.k_shot,
            hb.N, hb.D, hb.n_q_pts, hb.k_sub, _p(hb.nodes), hb.nodes.stride(0), _p(hb.Y), _p(hb.desc),
            _p(hb.assign), _p(hb.cluster_count), _p(hb.proto_ws), hb.proto_ws.numel(),
            HEAD_FPS_ONE_LAUNCH if hb.fps_one_launch else 0, _st()))


def label_propagate(hb, nbr, sigma, alpha=0.99, max_iter=200, tol=1e-6):
    assert nbr.shape == (1, hb.n_cap, hb.kp1) or nbr.shape == (hb.n_cap, hb.kp1)
    with _timed("label_propagate"):
        _lib.check(_lib.load().r3d_label_propagate(
            _p(hb.nodes), hb.nodes.stride(0), hb.D, _p(nbr), hb.kp1, _p(hb.Y), _p(hb.desc[HD_N_NODES:]),
            _p(hb.desc[HD_N_PROTO:]), hb.n_cap, float(sigma), float(alpha), int(max_iter), float(tol), _p(hb.Z),
            _p(hb.lp_ws), hb.lp_ws.numel(), _p(hb.stats), _st()))
    return hb.Z


def query_logits_ce(hb, n_q, n_classes, labels):
    dev = hb.Z.device
    logits = torch.empty(n_q, n_classes, hb.N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    pred = torch.empty(n_q, hb.N, device=dev, dtype=torch.int32)
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.is_contiguous()
    _lib.check(_lib.load().r3d_query_logits_ce(_p(hb.Z), _p(hb.desc[HD_N_PROTO:]), n_q, hb.N, n_classes,
                                               _p(labels), _p(logits), _p(loss), _p(pred), _st()))
    return logits, loss, pred


def clean_shot_detect(sfeat_pm, support_x, support_y, n_way, k_shot, N, want_debug=False):
    """shot_keep (n_way*k_shot) int32 of the eval-only clean-shot detection (mpti.py:178-223)."""
    M, ldf = _rows(sfeat_pm)
    S = n_way * k_shot
    assert M == S * N
    sx = support_x.reshape(S, -1, N).contiguous().float()
    sy = support_y.reshape(S, N).to(torch.int32).contiguous()
    dev = sfeat_pm.device
    keep = torch.empty(S, device=dev, dtype=torch.int32)
    dbg = torch.zeros(n_way, 2, 4 * k_shot, device=dev, dtype=torch.float32) if want_debug else None
    ws = torch.empty(_lib.load().r3d_clean_ws_words(n_way, k_shot), device=dev, dtype=torch.int32)
    _lib.check(_lib.load().r3d_clean_shot_detect(_p(sfeat_pm), ldf, sfeat_pm.shape[1], _p(sx), sx.shape[1], _p(sy),
                                                 n_way, k_shot, N, _p(keep), _p(dbg), _p(ws), _st()))
    return (keep, dbg) if want_debug else keep


def protonet_head(sfeat_pm, qfeat_pm, support_y, n_way, k_shot, N, method, scaler=10.0):
    """Similarity rows (n_q*N, 4) of the ProtoNet head; method 'cosine' | 'euclidean'."""
    M, ldf = _rows(sfeat_pm)
    Mq, ldq = _rows(qfeat_pm)
    codes = {"cosine": 0, "euclidean": 1}
    if method not in codes:
        raise NotImplementedError('Error! Distance computation method (%s) is unknown!' % method)
    sy = support_y.reshape(n_way * k_shot, N).to(torch.int32).contiguous()
    dev = sfeat_pm.device
    Z = torch.empty(Mq, 4, device=dev, dtype=torch.float32)
    ws = torch.empty(n_way * k_shot * 2 * 256, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().r3d_protonet_head(_p(sfeat_pm), ldf, _p(qfeat_pm), ldq, sfeat_pm.shape[1], _p(sy), n_way,
                                             k_shot, N, Mq, codes[method], float(scaler), _p(Z), _p(ws), _st()))
    return Z


def logits_ce_from_rows(Z, n_q, N, n_classes, labels):
    """Z (n_q*N, 4) -> logits (n_q, n_classes, N), CE loss, argmax."""
    dev = Z.device
    zero = torch.zeros(1, device=dev, dtype=torch.int32)
    logits = torch.empty(n_q, n_classes, N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    pred = torch.empty(n_q, N, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().r3d_query_logits_ce(_p(Z), _p(zero), n_q, N, n_classes, _p(labels), _p(logits), _p(loss),
                                               _p(pred), _st()))
    return logits, loss, pred


def miou_accumulate(pred, gt, lut, hist):
    """hist (3, n_classes) int64 += counts of this episode (eval_noise.py:39-62)."""
    pred = pred.to(torch.int32).contiguous()
    gt = gt.to(torch.int64).contiguous()
    _lib.check(_lib.load().r3d_miou_accumulate(_p(pred), _p(gt), pred.numel(), _p(lut), lut.numel(), hist.shape[1],
                                               _p(hist), _st()))
    return hist
