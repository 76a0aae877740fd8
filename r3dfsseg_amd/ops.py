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


class SegLayout:
    """Row layout of a batch of E episodes in the encoder (include/r3d.h, "BATCHES OF EPISODES"): the clouds of the
    batch are the rows of one matrix, episode after episode, [S support clouds | Q query clouds] each.  Segment
    2 e + p is getFeatures call p of episode e (models/mpti.py:434,436: support, then query) -- the unit of the
    BatchNorm batch statistics and the order in which the running statistics see them.  Q == 0: E equal segments
    of S clouds (a plain getFeatures call is E = 1, S = B)."""

    def __init__(self, E, S, Q, N):
        self.E, self.S, self.Q, self.N = E, S, Q, N
        self.clouds = S + Q                    # clouds per episode
        self.B = E * (S + Q)                   # clouds of the batch
        self.M = self.B * N                    # rows of the batch
        self.rows_a, self.rows_b = S * N, Q * N
        self.n_seg = E * (2 if Q else 1)
        self.ep_rows = (S + Q) * N             # rows between consecutive episodes

    def counts(self, per_row=1):
        """(count_a, count_b) of the BatchNorm statistics: elements per channel in a support / query segment."""
        return float(self.rows_a * per_row), float(self.rows_b * per_row)

    def aligned(self, tile=64):
        return self.rows_a % tile == 0 and self.rows_b % tile == 0


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


def is_point_major_view(x):
    """x (..., C, N): a transposed VIEW of a point-major buffer (..., N, C) -- what the reference's collate hands over
    (`torch.from_numpy(raw).transpose(2, 3)`, dataloaders/loader.py:1666,1679: the on-disk layout, never materialised)."""
    if x.dtype != torch.float32 or x.dim() < 3:
        return False
    C, N = x.shape[-2], x.shape[-1]
    return x.stride(-1) == C and x.stride(-2) == 1 and x.transpose(-1, -2).is_contiguous()


def input_layouts(x):
    """x (B, C, N) as the caller hands it -> (x_pm (B*N, C), x_cm (B, C, N) contiguous or None).
    A transposed view of point-major rows is used AS IT LIES: x_pm is that buffer (no transpose kernel, no copy) and
    x_cm is None, so the first kNN builds its channel-major operand from the point-major rows itself -- the read it
    does for every later layer anyway (r3d_knn_topk_batched packs x_pm when no channel-major copy is given).  A
    contiguous channel-major tensor goes through r3d_cm_to_pm and doubles as the first kNN's operand."""
    B, C, N = x.shape
    if is_point_major_view(x):
        return x.transpose(1, 2).reshape(B * N, C), None
    x = x.contiguous().float()
    return cm_to_pm(x), x


def cat_clouds(a, b, dim=1):
    """cat((a, b), dim) of cloud tensors (..., C, N) that keeps the point-major rows of point-major views (the result is
    again such a view: one contiguous copy of rows instead of a strided gather into channel-major)."""
    if is_point_major_view(a) and is_point_major_view(b):
        return torch.cat((a.transpose(-1, -2), b.transpose(-1, -2)), dim).transpose(-1, -2)
    return torch.cat((a, b), dim)


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


def knn(x_pm, B, N, k, mode=SCORE_DGCNN, n_valid=None, return_scores=False, x_cm=None, status=None, n_valid_stride=0):
    """x_pm (B*N, C) -> idx (B, N, k) int32, best first.  x_cm: optional (B, C, N) channel-major
    copy of the same points (saves the internal transpose of the streamed k <= 32 kernel).
    n_valid (device int32) with n_valid_stride > 0: set b has n_valid[b * n_valid_stride] valid rows."""
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
    # scratch for the bf16 pieces of the points: the streamed kernels' threshold pass then runs on the bf16 matrix core
    # (a lower bound is all it needs; neighbours and scores come from the fp32 pass: same bits either way)
    bf_ws, bf_words = None, 0
    if C % 64 == 0 and (k <= 32 or status is not None):
        bf_words = lib.r3d_knn_bf_ws_words(B, N, C)
        bf_ws = torch.empty(bf_words, device=dev, dtype=torch.float32)
    with _timed("knn_topk_l2" if mode == SCORE_L2 else "knn_topk"):
        _lib.check(lib.r3d_knn_topk_batched(_p(x_pm), ld, _p(x_cm), B, N, C, k, mode, _p(n_valid), n_valid_stride, _p(norm),
                                            _p(cm_ws), _p(idx), _p(sc), _p(status), _p(split_ws), split_words, _p(bf_ws),
                                            bf_words, _st()))
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


def attention(qkv, B, N, out, want_lse=False, group=0):
    """Inference attention of B clouds.  group > 0: the clouds are a batch of episodes of `group` clouds each; the
    key-axis split (and with it every output bit) is then the one of a single episode."""
    M, ld = _rows(qkv)
    M2, ldo = _rows(out)
    assert M == B * N and M2 == M and qkv.shape[1] == 192 and out.shape[1] == 64
    lib = _lib.load()
    lse = torch.empty(M, device=qkv.device, dtype=torch.float32)
    ws = torch.empty(lib.r3d_attention_ws_words_ep(B, N, group), device=qkv.device, dtype=torch.float32)
    with _timed("attention"):
        _lib.check(lib.r3d_attention_fwd_train_ep(_p(qkv), ld, B, N, _p(out), ldo, _p(lse), 0.0, ctypes.c_uint(0), None, group,
                                                  _p(ws), _st()))
    return lse if want_lse else None


class HeadBuffers:
    """Device buffers of the transductive heads of E episodes (capacity sized, no host sync).  E > 1: every array has a
    leading episode axis, episode e's system lives at [e] (nodes / Y / Z: rows [e n_cap, (e + 1) n_cap)).  E = 1, the
    single-episode head of the reference's schedule, keeps the plain shapes (desc (32,), stats (2,), ...)."""

    def __init__(self, n_way, k_shot, N, n_q_pts, k_sub, k_connect, D, device, E=1):
        lib = _lib.load()
        self.E = E
        self.n_way, self.k_shot, self.N, self.n_q_pts, self.k_sub, self.kp1, self.D = \
            n_way, k_shot, N, n_q_pts, k_sub, k_connect + 1, D
        # k_sub + 1 prototype slots per class: torch_cluster's float-rounded sample count gives k or k + 1 seeds
        # (csrc/head_proto.hip::hp_fps_count, models/mpti.py:612-613)
        self.n_cap = (n_way + 1) * (k_sub + 1) + n_q_pts
        assert lib.r3d_head_desc_words() == 32
        i32 = dict(device=device, dtype=torch.int32)
        f32 = dict(device=device, dtype=torch.float32)
        one = (lambda t: t[0]) if E == 1 else (lambda t: t)
        self.desc = one(torch.zeros(E, 32, **i32))
        self.nodes = torch.empty(E * self.n_cap, D, **f32)
        # label columns travel as float4 per node.  More than 3 ways (5..8 classes): TWO planes of 4 columns, plane 1
        # (classes 4..7) behind the E systems of plane 0 -- the label propagation is column-wise independent and solves
        # the planes one after the other on the same graph (csrc/head_graph.hip)
        self.planes = 1 if n_way <= 3 else 2
        self.Y = torch.empty(self.planes * E * self.n_cap, 4, **f32)
        self.Z = torch.empty(self.planes * E * self.n_cap, 4, **f32)
        self.proto_words = lib.r3d_head_proto_ws_words(n_way, k_shot, N)
        self.proto_stride = (self.proto_words + 3) // 4 * 4      # even (64-bit words inside), 16-byte rows
        self.proto_ws = one(torch.empty(E, self.proto_stride, **i32))
        self.lp_words = lib.r3d_lp_ws_words(self.n_cap, self.kp1)
        self.lp_stride = (self.lp_words + 3) // 4 * 4             # float4 arrays inside
        self.lp_ws = one(torch.empty(E, self.lp_stride, **i32))
        self.assign = one(torch.empty(E, 2 * n_way * k_shot * N, **i32))
        self.cluster_count = one(torch.zeros(E, self.n_cap, **i32))
        self.stats = one(torch.zeros(E, 2, **i32))                # per system: {converged, CG iterations}
        self.stats2 = torch.zeros(E, 2, **i32) if self.planes == 2 else None   # ... of the second plane's solves
        self.knn_status = torch.zeros(1, **i32)                   # ONE word for the batch (bit 0: survivor overflow)
        self.stats_bwd = one(torch.zeros(E, 2, **i32))
        # all FPS rounds in one persistent launch: its workgroups that hold points must be co-resident (~500 slots of
        # this kernel at D <= 192, 250 above); fps_group episodes share a launch, fps_slots caps what may be resident
        # (the library clamps the group to the kernel's real occupancy as well: csrc/head_proto.hip::fps_slots_clamp)
        self.fps_one_launch = True
        self.fps_blocks = (n_way * k_shot * N + 255) // 256 + n_way + 1
        self.fps_slots = 500 if D <= 192 else 250
        off = (ctypes.c_long * 6)()
        lib.r3d_head_proto_ws_offsets(n_way, k_shot, N, off)
        self.ws_off = list(off)
        lib.r3d_lp_ws_offsets(self.n_cap, self.kp1, off)
        self.lp_off = dict(zip(("row_ptr", "col", "val", "dinv", "agg", "cg"), off))

    @property
    def fps_group(self):
        return max(1, self.fps_slots // self.fps_blocks)

    def n_nodes_ptr(self):
        return self.desc.view(-1)[HD_N_NODES:]   # episode e's count 32 words further on

    def n_proto_ptr(self):
        return self.desc.view(-1)[HD_N_PROTO:]

    def csr(self, e=0):
        """(n, row_ptr (n+1) int64, col (nnz) int64, val (nnz) fp32) of the normalised graph S the last
        r3d_label_propagate left in episode e's workspace (synchronises; tests, tools and bench.py's byte counts)."""
        n = int(self.desc.view(-1, 32)[e, HD_N_NODES].item())
        o = self.lp_off
        ws = self.lp_ws.view(-1, self.lp_stride)[e]
        row_ptr = ws[o["row_ptr"]:o["row_ptr"] + n + 1].to(torch.int64)
        nnz = int(row_ptr[-1].item())
        col = ws[o["col"]:o["col"] + (nnz + 1) // 2].view(torch.int16)[:nnz].to(torch.int64) & 0xffff
        val = ws[o["val"]:o["val"] + nnz].view(torch.float32)
        return n, row_ptr, col, val


def head_prototypes(hb, support_y, shot_keep, sfeat_pm, qfeat_pm, feat_ep_rows=0):
    """Prototypes + node matrices of hb.E episodes.  sfeat_pm / qfeat_pm: the support / query rows of episode 0 inside
    the batch's feature matrix; episode e's rows start feat_ep_rows rows further on.  support_y (E, S, N) int32,
    shot_keep optional (E, S) int32."""
    S = hb.n_way * hb.k_shot
    E = hb.E
    ldf, ldq = sfeat_pm.stride(0), qfeat_pm.stride(0)
    assert sfeat_pm.dtype == torch.float32 and qfeat_pm.dtype == torch.float32 and sfeat_pm.stride(1) == 1
    assert E == 1 or feat_ep_rows >= S * hb.N
    assert support_y.dtype == torch.int32 and support_y.is_contiguous() and support_y.numel() == E * S * hb.N
    assert shot_keep is None or (shot_keep.dtype == torch.int32 and shot_keep.numel() == E * S)
    with _timed("head_prototypes"):
        _lib.check(_lib.load().r3d_head_prototypes_batched(
            E, hb.fps_group, _p(support_y), S * hb.N, _p(shot_keep), S, _p(sfeat_pm), ldf, feat_ep_rows, _p(qfeat_pm), ldq,
            feat_ep_rows, hb.n_way, hb.k_shot, hb.N, hb.D, hb.n_q_pts, hb.k_sub, _p(hb.nodes), hb.nodes.stride(0), hb.n_cap,
            _p(hb.Y), _p(hb.desc), 32, _p(hb.assign), 2 * S * hb.N, _p(hb.cluster_count), hb.n_cap, _p(hb.proto_ws),
            hb.proto_words, hb.proto_stride, HEAD_FPS_ONE_LAUNCH if hb.fps_one_launch else 0, _st()))


def knn_nodes(hb, exact=False):
    """201-NN lists (E, n_cap, kp1) of every episode's graph nodes (mpti.py:731-736).  exact: the insertion kernel
    (always exact); otherwise the append-and-rank kernel, whose survivor-buffer overflow sets hb.knn_status."""
    nbr = knn(hb.nodes, hb.E, hb.n_cap, hb.kp1, mode=SCORE_L2, n_valid=hb.n_nodes_ptr(), n_valid_stride=32,
              status=None if exact else hb.knn_status)
    if exact:
        hb.knn_status.zero_()
    return nbr


def label_propagate(hb, nbr, sigma, alpha=0.99, max_iter=200, tol=1e-6):
    assert nbr.numel() == hb.E * hb.n_cap * hb.kp1 and nbr.is_contiguous()
    with _timed("label_propagate"):
        _lib.check(_lib.load().r3d_label_propagate_batched(
            hb.E, _p(hb.nodes), hb.nodes.stride(0), hb.D, _p(nbr), hb.kp1, _p(hb.Y), _p(hb.n_nodes_ptr()),
            _p(hb.n_proto_ptr()), 32, hb.n_cap, float(sigma), float(alpha), int(max_iter), float(tol), _p(hb.Z), _p(hb.lp_ws),
            hb.lp_words, hb.lp_stride, _p(hb.stats), 2, _st()))
        if hb.planes == 2:  # label columns 4..7: the same systems, further right-hand sides
            pl = hb.E * hb.n_cap
            _lib.check(_lib.load().r3d_label_propagate_solve_batched(
                hb.E, _p(hb.Y[pl:]), _p(hb.n_nodes_ptr()), 32, hb.n_cap, hb.kp1, float(alpha), int(max_iter), float(tol),
                _p(hb.Z[pl:]), _p(hb.lp_ws), hb.lp_words, hb.lp_stride, _p(hb.stats2), 2, _st()))
            # one {converged, iterations} pair per system for the callers: converged = both, iterations = the larger
            st1 = hb.stats.view(hb.E, 2)
            st1[:, 0] = torch.minimum(st1[:, 0], hb.stats2[:, 0])
            st1[:, 1] = torch.maximum(st1[:, 1], hb.stats2[:, 1])
    return hb.Z


def graph_weights_verify(hb, sigma):
    """Recompute the gaussian edge weights of the graphs the last label_propagate left in hb (call with the chip otherwise
    idle) and return the number of entries whose bits differ from the ones the solve used (device int32 tensor)."""
    lib = _lib.load()
    dev = hb.nodes.device
    scratch = torch.empty(hb.E * lib.r3d_graph_weights_verify_words(hb.n_cap, hb.kp1), device=dev, dtype=torch.float32)
    bad = torch.zeros(1, device=dev, dtype=torch.int32)
    _lib.check(lib.r3d_graph_weights_verify(hb.E, _p(hb.nodes), hb.nodes.stride(0), hb.D, _p(hb.n_nodes_ptr()), 32, hb.n_cap,
                                            hb.kp1, float(sigma), _p(hb.lp_ws), hb.lp_words, hb.lp_stride, _p(scratch), _p(bad),
                                            _st()))
    return bad


def query_logits_ce(hb, n_q, n_classes, labels):
    """labels (E, n_q, N) int64 or None -> logits (E, n_q, n_classes, N), loss (E,), pred (E, n_q, N) int32 (E = 1: without
    the episode axis)."""
    dev = hb.Z.device
    E = hb.E
    logits = torch.empty(E, n_q, n_classes, hb.N, device=dev, dtype=torch.float32)
    loss = torch.empty(E, device=dev, dtype=torch.float32)
    pred = torch.empty(E, n_q, hb.N, device=dev, dtype=torch.int32)
    if labels is not None:
        assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == E * n_q * hb.N
    _lib.check(_lib.load().r3d_query_logits_ce_batched(E, _p(hb.Z), hb.n_cap, _p(hb.n_proto_ptr()), 32, n_q, hb.N, n_classes,
                                                       _p(labels), _p(logits), _p(loss), _p(pred), _st()))
    if E == 1:  # the single-episode shapes: (n_q, n_classes, N), 0-d, (n_q, N)
        return logits[0], loss[0], pred[0]
    return logits, loss, pred


def clean_shot_detect(sfeat_pm, support_x, support_y, n_way, k_shot, N, want_debug=False, E=1, feat_ep_rows=0):
    """shot_keep (E, n_way*k_shot) int32 of the eval-only clean-shot detection (mpti.py:178-223).  sfeat_pm: support rows
    of episode 0 inside the batch's feature matrix, episode e feat_ep_rows rows further on; support_x (E, S, Cin, N)."""
    ldf = sfeat_pm.stride(0)
    S = n_way * k_shot
    sx = support_x.reshape(E * S, -1, N).contiguous().float()
    sy = support_y.reshape(E * S, N).to(torch.int32).contiguous()
    dev = sfeat_pm.device
    keep = torch.empty(E, S, device=dev, dtype=torch.int32)
    dbg = torch.zeros(E, n_way, 2, 4 * k_shot, device=dev, dtype=torch.float32) if want_debug else None
    words = _lib.load().r3d_clean_ws_words(n_way, k_shot)
    ws = torch.empty(E, words, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().r3d_clean_shot_detect_batched(E, _p(sfeat_pm), ldf, feat_ep_rows, sfeat_pm.shape[1], _p(sx),
                                                         sx.shape[1], _p(sy), n_way, k_shot, N, _p(keep), _p(dbg), _p(ws),
                                                         words, _st()))
    if E == 1:
        keep = keep[0]
        dbg = dbg[0] if dbg is not None else None
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
    Z = torch.empty((1 if n_way <= 3 else 2) * Mq, 4, device=dev, dtype=torch.float32)  # (planes of 4 classes)
    ws = torch.empty(n_way * k_shot * 2 * 256, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().r3d_protonet_head(_p(sfeat_pm), ldf, _p(qfeat_pm), ldq, sfeat_pm.shape[1], _p(sy), n_way,
                                             k_shot, N, Mq, codes[method], float(scaler), _p(Z), _p(ws), _st()))
    return Z


def logits_ce_from_rows(Z, n_q, N, n_classes, labels):
    """Z (n_q*N, 4) -- (2, n_q*N, 4) for more than 4 classes -- -> logits (n_q, n_classes, N), CE loss, argmax."""
    dev = Z.device
    zero = torch.zeros(1, device=dev, dtype=torch.int32)
    logits = torch.empty(n_q, n_classes, N, device=dev, dtype=torch.float32)
    loss = torch.empty((), device=dev, dtype=torch.float32)
    pred = torch.empty(n_q, N, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().r3d_query_logits_ce_batched(1, _p(Z), n_q * N, _p(zero), 0, n_q, N, n_classes, _p(labels),
                                                       _p(logits), _p(loss.view(1)), _p(pred), _st()))
    return logits, loss, pred


def miou_accumulate(pred, gt, lut, hist):
    """hist (3, n_classes) int64 += counts of this episode (eval_noise.py:39-62)."""
    pred = pred.to(torch.int32).contiguous()
    gt = gt.to(torch.int64).contiguous()
    _lib.check(_lib.load().r3d_miou_accumulate(_p(pred), _p(gt), pred.numel(), _p(lut), lut.numel(), hist.shape[1],
                                               _p(hist), _st()))
    return hist
