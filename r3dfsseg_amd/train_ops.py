"""Training-mode launch orchestration (forward with batch-statistics BatchNorm + backward).

Two torch.autograd.Function objects wrap the HIP kernels so that the reference's
``loss.backward(); optimizer.step()`` (models/mpti_learner.py:68-70) works unchanged:
``EncoderTrainFn`` (DGCNN + BaseLearner + SelfAttention, i.e. getFeatures, models/mpti.py:579-589)
and, in head_train.py, the transductive head with its losses.  Nothing here computes: every tensor
operation is a call into libr3d_hip.so; torch provides memory, the autograd graph edge and Adam.

Everything works on a BATCH OF E EPISODES laid out as ops.SegLayout describes (E = 1: the reference's schedule, one
episode per step): kNN, every GEMM, the attention and the edge passes are ONE launch over all E (S + Q) clouds, while
every BatchNorm keeps the statistics of the 2 E getFeatures calls apart (segments; models/mpti.py:434,436) and updates /
records the running statistics in the reference's order -- episode after episode, support call then query call.  A
segment's statistics, and with them every per-episode result of the forward pass, are bit for bit the same whether
the episode runs alone or inside a batch (the kernels partition their reductions by the segment, not by the batch).
"""
import ctypes

import torch

from . import _lib, ops
from .ops import SegLayout, _p, _rows, _st, _timed

BN_EPS, BN_MOM = 1e-5, 0.1
# False while a training forward must leave the BatchNorm running statistics alone
update_running_stats = True


class BNRecorder:
    """Batch statistics of launch sequences whose running-statistics update is deferred: a captured episode that may be
    in flight beside others, or a batched step that is only kept once its solver status is known.  A sequence RECORDS
    (batch mean, unbiased variance) of every BatchNorm call -- record index_dev + 2 e + p for call p (0 support, 1 query;
    mpti.py:434,436) of the sequence's episode e -- and `apply()` folds the records into the running statistics in
    exactly that order: bit for bit what the reference's one-episode-at-a-time schedule gives."""

    def __init__(self, max_episodes, device):
        self.max_records = 2 * max_episodes
        self.device = device
        self.tables = {}        # BatchNorm module -> (records (max_records, 2, C), conv bias or None)
        self.index_dev = None   # int32 device word: first record of the running launch sequence (None: 0)

    def slot_for(self, bnmod, bias):
        if bnmod not in self.tables:
            rec = torch.zeros(self.max_records, 2, bnmod.num_features, device=self.device, dtype=torch.float32)
            self.tables[bnmod] = (rec, bias.detach() if bias is not None else None)
        return self.tables[bnmod][0]

    def apply(self, n_episodes):
        lib = _lib.load()
        assert 2 * n_episodes <= self.max_records
        with torch.no_grad():
            for bnmod, (rec, bias) in self.tables.items():
                C = bnmod.num_features
                _lib.check(lib.r3d_bn_running_update(_p(rec), 2 * n_episodes, 2 * C, C, BN_MOM, _p(bias),
                                                     _p(bnmod.running_mean), _p(bnmod.running_var), _st()))
                bnmod.num_batches_tracked += 2 * n_episodes


bn_recorder = None  # set by the owner of a deferred sequence (episode_graph.EpisodeGraphs, batched.EpisodeBatchRunner)


class deferred_running_stats:
    """``with deferred_running_stats(model) as rec: <training forward of ONE episode>`` records the episode's BatchNorm batch
    statistics instead of updating the running statistics; ``rec.apply(1)`` afterwards folds them in.  The eager
    one-episode schedule runs an attempt that may be discarded (CG launch budget, 201-NN overflow, FPS time-out) and
    redone on the conservative schedule: only the attempt that is kept may count in the running statistics."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        global bn_recorder
        rec = self.model.__dict__.get("_bn_rec1")
        if rec is None:
            rec = self.model.__dict__["_bn_rec1"] = BNRecorder(1, next(self.model.parameters()).device)
        self.saved, bn_recorder = bn_recorder, rec
        return rec

    def __exit__(self, *exc):
        global bn_recorder
        bn_recorder = self.saved
        return False


def _f(n, dev):
    return torch.empty(n, device=dev, dtype=torch.float32)


class BNVec:
    """BatchNorm vectors of every segment: table (n_seg, 4, C) = scale | shift | mean | invstd; the kernels address
    segment s at (pointer + s * stride)."""

    def __init__(self, n_seg, C, dev):
        self.t = torch.empty(n_seg, 4, C, device=dev, dtype=torch.float32)
        self.stride = 4 * C

    @property
    def scale(self): return self.t[:, 0]
    @property
    def shift(self): return self.t[:, 1]
    @property
    def mean(self): return self.t[:, 2]
    @property
    def invstd(self): return self.t[:, 3]

    def ptrs(self):
        return _p(self.t[0, 0]), _p(self.t[0, 1]), _p(self.t[0, 2]), _p(self.t[0, 3])


# ----------------------------------------------------------------------------- thin wrappers
def colstats(X, C, seg, mode=0, DY=None, bn=None, act=0):
    """Per-segment column sums (n_seg, 2, C): mode 0 (sum x, sum x^2); mode 1 (sum du, sum du zhat) of the BN backward."""
    M, ldx = _rows(X)
    dev = X.device
    lib = _lib.load()
    sums = _f(seg.n_seg * 2 * C, dev).view(seg.n_seg, 2, C)
    ws = _f(lib.r3d_colstats_seg_ws_words(M, C, seg.rows_a, seg.rows_b), dev)
    lddy = DY.stride(0) if DY is not None else 0
    sc, sh, mu, is_ = bn.ptrs() if bn is not None else (None, None, None, None)
    with _timed("bn_stats"):
        _lib.check(lib.r3d_colstats_seg(_p(X), ldx, _p(DY), lddy, M, C, seg.rows_a, seg.rows_b, mode, sc, sh, mu, is_,
                                        bn.stride if bn is not None else 0, act, _p(sums), _p(ws), _st()))
    return sums


def bn_fold(sums, counts, bnmod, bias=None):
    """Per-segment batch mean / invstd -> BNVec; the module's running statistics are updated like nn.BatchNorm in train
    mode, segment after segment (or recorded for a deferred update, see BNRecorder).  `bias`: conv bias folded away by the
    mean subtraction (it only shifts the running mean)."""
    n_seg, _, C = sums.shape
    dev = sums.device
    lib = _lib.load()
    bn = BNVec(n_seg, C, dev)
    rec = bn_recorder
    if rec is not None:
        table, idx_dev = rec.slot_for(bnmod, bias), rec.index_dev
    elif update_running_stats:  # immediate update: the same records, applied right behind the fold
        table, idx_dev = torch.empty(n_seg, 2, C, device=dev, dtype=torch.float32), None
    else:
        table, idx_dev = None, None
    sc, sh, mu, is_ = bn.ptrs()
    _lib.check(lib.r3d_bn_fold_seg(_p(sums), n_seg, counts[0], counts[1], C, _p(bnmod.weight), _p(bnmod.bias), BN_EPS, BN_MOM,
                                   None, None, mu, is_, sc, sh, bn.stride, _p(table), _p(idx_dev), 2 * C, _st()))
    if rec is None and table is not None:
        with torch.no_grad():
            _lib.check(lib.r3d_bn_running_update(_p(table), n_seg, 2 * C, C, BN_MOM, _p(bias.detach()) if bias is not None else None,
                                                 _p(bnmod.running_mean), _p(bnmod.running_var), _st()))
            bnmod.num_batches_tracked += n_seg
    return bn


def affine_act(Z, bn, act, seg, out=None):
    M, ldz = _rows(Z)
    C = Z.shape[1]
    if out is None:
        out = torch.empty(M, C, device=Z.device, dtype=torch.float32)
    sc, sh, _, _ = bn.ptrs()
    _lib.check(_lib.load().r3d_affine_act_seg(_p(Z), ldz, M, C, seg.rows_a, seg.rows_b, sc, sh, bn.stride, act, _p(out),
                                              out.stride(0), _st()))
    return out


def bn_bwd_apply(Z, DY, bn, act, sums, counts, seg, out=None):
    M, ldz = _rows(Z)
    C = Z.shape[1]
    DZ = torch.empty(M, C, device=Z.device, dtype=torch.float32) if out is None else out
    sc, sh, mu, is_ = bn.ptrs()
    _lib.check(_lib.load().r3d_bn_bwd_apply_seg(_p(Z), ldz, _p(DY), DY.stride(0), M, C, seg.rows_a, seg.rows_b, sc, sh, mu, is_,
                                                bn.stride, act, _p(sums), counts[0], counts[1], _p(DZ), DZ.stride(0), _st()))
    return DZ


def gemm_tn(A, B):
    """A^T B: (M, Ca), (M, Cb) -> (Ca, Cb)."""
    M, lda = _rows(A)
    M2, ldb = _rows(B)
    assert M == M2
    Ca, Cb = A.shape[1], B.shape[1]
    lib = _lib.load()
    out = torch.empty(Ca, Cb, device=A.device, dtype=torch.float32)
    ws = _f(lib.r3d_gemm_tn_ws_words(M, Ca, Cb), A.device)
    with _timed("gemm_tn"):
        _lib.check(lib.r3d_gemm_tn(_p(A), lda, _p(B), ldb, M, Ca, Cb, 1.0, _p(out), 0, _p(ws), _st()))
    return out


def conv_acc(X, W, out):
    """out += X W^T."""
    M, ldx = _rows(X)
    with _timed("pointwise_conv"):
        _lib.check(_lib.load().r3d_pointwise_conv_acc(_p(X), ldx, _p(W), M, X.shape[1], W.shape[0], None, None, 0, _p(out),
                                                      out.stride(0), _st()))


def add_cols(src, dst):
    M, lds = _rows(src)
    _lib.check(_lib.load().r3d_add_cols(_p(src), lds, _p(dst), dst.stride(0), M, src.shape[1], _st()))


def _one_segment(M):
    return SegLayout(1, 1, 0, M)  # (rows_a = M: a matrix normalised as a whole)


# ----------------------------------------------------------------------------- conv + BN + act layer
def conv_bn_fwd(X, W2d, bnmod, act, bias=None, out=None, seg=None):
    """Returns (y, saved).  `seg`: the row segments that are normalised separately (ops.SegLayout; None: the whole
    matrix is one batch); the GEMM runs once over all rows."""
    # raw z (a conv bias cancels under batch statistics) and its column sums from the same GEMM launch
    M, ldx = _rows(X)
    C = W2d.shape[0]
    lib = _lib.load()
    seg = _one_segment(M) if seg is None else seg
    assert seg.M == M
    z = torch.empty(M, C, device=X.device, dtype=torch.float32)
    if seg.aligned(64):  # (also for one segment: a segment's partition must not depend on what it is batched with)
        sums = _f(seg.n_seg * 2 * C, X.device).view(seg.n_seg, 2, C)
        ws = _f(lib.r3d_pointwise_conv_stats_ws_words(M, C), X.device)
        with _timed("pointwise_conv"):
            _lib.check(lib.r3d_pointwise_conv_stats_seg(_p(X), ldx, _p(W2d), M, X.shape[1], C, _p(z), C, seg.rows_a, seg.rows_b,
                                                        _p(sums), _p(ws), _st()))
    else:  # segments that do not end on the GEMM's 64-row tiles: the statistics take their own pass over z
        ops.pointwise_conv(X, W2d, out=z)
        sums = colstats(z, C, seg, mode=0)
    bn = bn_fold(sums, seg.counts(), bnmod, bias)
    y = affine_act(z, bn, act, seg, out=out)
    return y, (X, W2d, z, bn, act, seg)


def conv_bn_bwd(saved, dY, want_dx=True, dx_acc=None):
    """Returns (dW, dgamma, dbeta, dbias, dX), summed over the segments.  dX is accumulated into dx_acc when given."""
    X, W2d, z, bn, act, seg = saved
    C = W2d.shape[0]
    M = z.shape[0]
    sums = colstats(z, C, seg, mode=1, DY=dY, bn=bn, act=act)
    dz = bn_bwd_apply(z, dY, bn, act, sums, seg.counts(), seg)
    dW = gemm_tn(dz, X)  # one launch over all rows: the segments' (episodes') weight gradients add
    tot = sums.sum(0) if seg.n_seg > 1 else sums[0]
    # a conv bias in front of a training-mode BatchNorm has gradient sum_m dz = 0 identically (dz is the
    # BN backward output, whose column sums vanish); the reference's autograd returns round-off noise there
    dbias = torch.zeros(C, device=z.device, dtype=torch.float32)
    dX = None
    if want_dx:
        Wt = W2d.t().contiguous()
        if dx_acc is not None:
            conv_acc(dz, Wt, dx_acc)
        else:
            dX = ops.pointwise_conv(dz, Wt)
    return dW, tot[1], tot[0], dbias, dX


# ----------------------------------------------------------------------------- EdgeConv layer
def edgeconv_train_fwd(inp, idx, ec, B, N, out, seg=None):
    """`seg`: the cloud segments whose BatchNorm statistics stay apart (None: all B clouds are one batch); the PQ GEMM and
    every edge pass run once over all clouds."""
    lib = _lib.load()
    dev = inp.device
    seg = SegLayout(1, B, 0, N) if seg is None else seg
    assert seg.B == B and seg.N == N
    W1 = ec.layer[0].weight.reshape(64, -1)
    C = W1.shape[1] // 2
    Wpq = torch.cat((W1[:, :C], W1[:, C:] - W1[:, :C]), 0).contiguous()
    PQ = ops.pointwise_conv(inp, Wpq)
    K = idx.shape[-1]
    ws = _f(lib.r3d_edgeconv_train_ws_words(B, N), dev)
    W2 = ec.layer[3].weight.reshape(64, 64).contiguous()
    M = B * N
    argmax = torch.empty(M, 64, device=dev, dtype=torch.int32)
    argmin = torch.empty(M, 64, device=dev, dtype=torch.int32)
    zmax = torch.empty(M, 64, device=dev, dtype=torch.float32)
    zmin = torch.empty(M, 64, device=dev, dtype=torch.float32)
    idx3 = idx.view(B, N, K)
    sums1 = _f(seg.n_seg * 128, dev).view(seg.n_seg, 2, 64)
    sums2 = _f(seg.n_seg * 128, dev).view(seg.n_seg, 2, 64)
    counts = seg.counts(per_row=K)  # edges per segment
    esum = torch.empty(M, 64, device=dev, dtype=torch.float32)  # sum_t e1 per point, for the backward
    with _timed("edgeconv"):
        _lib.check(lib.r3d_edge_stats1(_p(PQ), _p(idx3), B, N, K, seg.S, seg.Q, _p(sums1), _p(esum), _p(ws), _st()))
    bn1 = bn_fold(sums1, counts, ec.layer[1])
    sc1, sh1, _, _ = bn1.ptrs()
    with _timed("edgeconv"):  # ONE edge-GEMM pass: z2 statistics and per-point max / min of z2
        _lib.check(lib.r3d_edgeconv_train_fwd_minmax(_p(PQ), _p(idx3), sc1, sh1, bn1.stride, _p(W2), B, N, K, seg.S, seg.Q,
                                                     _p(zmax), _p(zmin), _p(argmax), _p(argmin), _p(sums2), _p(ws), _st()))
    bn2 = bn_fold(sums2, counts, ec.layer[4])
    sc2, sh2, _, _ = bn2.ptrs()
    with _timed("edgeconv"):  # BN2 + LeakyReLU is monotone per channel: pick max or min, in place
        _lib.check(lib.r3d_edge_select(_p(zmax), _p(zmin), _p(argmax), _p(argmin), sc2, sh2, bn2.stride, M, seg.rows_a,
                                       seg.rows_b, _p(out), out.stride(0), _st()))
    return (inp, idx3, Wpq, PQ, W2, bn1, bn2, argmax, zmax, C, seg, esum)


def edgeconv_train_bwd(saved, dout, B, N, dx_acc):
    """Returns (dW1 (64,2C,1,1), dg1, db1, dW2 (64,64,1,1), dg2, db2); input gradient accumulated into dx_acc."""
    inp, idx3, Wpq, PQ, W2, bn1, bn2, argmax, zmax, C, seg, esum = saved
    lib = _lib.load()
    dev = PQ.device
    K = idx3.shape[-1]
    M = B * N
    dPQ = _f(M * 128, dev).view(M, 128)
    ws = _f(lib.r3d_edgeconv_train_ws_words(B, N), dev)
    # the reverse neighbour list of ALL clouds in one launch: the input gradient is a gather over incoming edges
    # (deterministic, no float atomics)
    rev = torch.empty(lib.r3d_edge_reverse_ws_words(B, N, K), device=dev, dtype=torch.int32)
    with _timed("edgeconv_bwd"):
        _lib.check(lib.r3d_edge_reverse(_p(idx3), B, N, K, _p(rev), rev.numel(), _st()))
    bn2_sums = colstats(zmax, 64, seg, mode=1, DY=dout, bn=bn2, act=ops.ACT_LRELU)
    DY1, BE = _f(M * K * 64, dev), _f(M * 128, dev)
    dW2, bn1_sums = _f(64 * 64, dev), _f(seg.n_seg * 128, dev).view(seg.n_seg, 2, 64)
    s1, t1, m1, i1 = bn1.ptrs()
    s2, t2, m2, i2 = bn2.ptrs()
    with _timed("edgeconv_bwd"):
        _lib.check(lib.r3d_edgeconv_bwd(_p(PQ), _p(idx3), s1, t1, m1, i1, _p(W2), s2, t2, m2, i2, bn1.stride, _p(bn2_sums),
                                        _p(dout), dout.stride(0), _p(argmax), _p(zmax), _p(esum), B, N, K, seg.S, seg.Q, _p(DY1),
                                        _p(BE), _p(rev),
                                        _p(dW2), _p(bn1_sums), _p(dPQ), _p(ws), _st()))
    t1s = bn1_sums.sum(0) if seg.n_seg > 1 else bn1_sums[0]
    t2s = bn2_sums.sum(0) if seg.n_seg > 1 else bn2_sums[0]
    dWpq = gemm_tn(dPQ, inp)  # (128, C): rows 0..63 = dP^T x, rows 64..127 = dQ^T x
    dW1 = torch.cat((dWpq[:64] - dWpq[64:], dWpq[64:]), 1).reshape(64, 2 * C, 1, 1)
    if dx_acc is not None:
        conv_acc(dPQ, Wpq.t().contiguous(), dx_acc)
    return dW1, t1s[1], t1s[0], dW2.view(64, 64, 1, 1), t2s[1], t2s[0]


# ----------------------------------------------------------------------------- encoder
class EncoderTrainFn(torch.autograd.Function):
    """getFeatures in training mode.  forward(x (B,C_in,N), model, seed, *params) -> feat (B*N, 192).

    ``ctx.seg`` (ops.SegLayout, set by the caller before forward; default: the B clouds are one getFeatures call) says
    which clouds are which episode's support / query call.  The two getFeatures calls of every episode of the batch
    (mpti.py:434,436) go through ONE launch sequence: kNN, every GEMM and the attention see all clouds in one grid,
    every BatchNorm keeps the statistics of the calls apart (and updates / records them in the reference's order), the
    attention dropout of episode e draws the mask of seed + 2 e, and backward returns the SUM of all calls' parameter
    gradients."""

    @staticmethod
    def forward(ctx, x, model, seed, *params):
        enc, base, att = model.encoder, model.base_learner, model.att_learner
        lib = _lib.load()
        B, _, N = x.shape
        M = B * N
        dev = x.device
        seg = getattr(ctx, "seg", None)
        if seg is None:
            seg = SegLayout(1, B, 0, N)
        assert seg.B == B and seg.N == N
        x_pm, x_cm = ops.input_layouts(x)
        cat = torch.empty(M, 64 * enc.n_edgeconv, device=dev, dtype=torch.float32)
        inp, ec_saved = x_pm, []
        for l in range(enc.n_edgeconv):
            idx = ops.knn(inp, B, N, enc.k, x_cm=x_cm if l == 0 else None)
            if enc.idx_patch is not None:  # parity tests only (dgcnn.DGCNN.idx_patch)
                idx = enc.idx_patch(l, idx.view(B, N, enc.k)).view(idx.shape)
            out = cat[:, 64 * l:64 * (l + 1)]
            ec_saved.append(edgeconv_train_fwd(inp, idx, enc.edge_convs[l], B, N, out, seg))
            inp = out
        h, mlp_saved = cat, []
        for jn in range(len(enc.conv.layer_dims)):
            W = enc.conv.layer[3 * jn].weight
            h, sv = conv_bn_fwd(h, W.reshape(W.shape[0], -1).contiguous(), enc.conv.layer[3 * jn + 1], ops.ACT_LRELU, seg=seg)
            mlp_saved.append(sv)
        level2 = h
        feat = torch.empty(M, model.feat_dim, device=dev, dtype=torch.float32)
        ops.copy_cols(cat[:, :64], feat[:, :64])
        hb, base_saved = level2, []
        for i, seq in enumerate(base.convs):
            last = i == base.num_convs - 1
            W = seq[0].weight
            hb, sv = conv_bn_fwd(hb, W.reshape(W.shape[0], -1).contiguous(), seq[1], ops.ACT_NONE if last else ops.ACT_RELU,
                                 bias=seq[0].bias, out=feat[:, 128:] if last else None, seg=seg)
            base_saved.append(sv)
        Wqkv, qscale = att._fold()
        qkv = ops.pointwise_conv(level2, Wqkv, qscale, None, ops.ACT_NONE)
        lse = torch.empty(M, device=dev, dtype=torch.float32)
        p_drop = float(att.dropout.p)
        aws = _f(lib.r3d_attention_ws_words_ep(B, N, seg.clouds), dev)
        with _timed("attention"):
            _lib.check(lib.r3d_attention_fwd_train_ep(_p(qkv), 192, B, N, _p(feat[:, 64:128]), feat.stride(0), _p(lse), p_drop,
                                                      ctypes.c_uint(seed & 0xffffffff), _p(model._slot.seed_dev), seg.clouds,
                                                      _p(aws), _st()))
        ctx.model, ctx.dims, ctx.seed_dev, ctx.seg = model, (B, N, seed, p_drop), model._slot.seed_dev, seg
        model._dbg_idx = [sv[1] for sv in ec_saved]  # neighbour lists of this pass (parity tests inject them into the oracle)
        if getattr(model, "_trace", None) is not None:  # one entry per getFeatures call
            for e in range(seg.E):
                for b0, Bs in ((e * seg.clouds, seg.S), (e * seg.clouds + seg.S, seg.Q)):
                    if Bs:
                        model._trace.setdefault("idx", []).append([sv[1][b0:b0 + Bs] for sv in ec_saved])
                        model._trace.setdefault("argmax", []).append([sv[7][b0 * N:(b0 + Bs) * N] for sv in ec_saved])
        ctx.saved = (ec_saved, mlp_saved, base_saved, cat, level2, Wqkv, qkv, lse, feat, aws)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model = ctx.model
        enc, base, att = model.encoder, model.base_learner, model.att_learner
        B, N, seed, p_drop = ctx.dims
        seg = ctx.seg
        ec_saved, mlp_saved, base_saved, cat, level2, Wqkv, qkv, lse, feat, aws = ctx.saved
        lib = _lib.load()
        dev = dfeat.device
        M = B * N
        dfeat = dfeat.contiguous()
        class _G(dict):  # gradients keyed by parameter identity
            def __setitem__(self, k, v):
                dict.__setitem__(self, id(k), v)
        g = _G()
        dlevel2 = torch.zeros(M, level2.shape[1], device=dev, dtype=torch.float32)
        # --- BaseLearner (mpti.py:35-40)
        d = dfeat[:, 128:]
        for i in reversed(range(base.num_convs)):
            seq = base.convs[i]
            dW, dg, db, dbias, dX = conv_bn_bwd(base_saved[i], d, want_dx=True, dx_acc=dlevel2 if i == 0 else None)
            g[seq[0].weight] = dW.view_as(seq[0].weight)
            g[seq[0].bias] = dbias
            g[seq[1].weight], g[seq[1].bias] = dg, db
            d = dX
        # --- SelfAttention (attention.py:39-46)
        dqkv = torch.empty(M, 192, device=dev, dtype=torch.float32)
        with _timed("attention_bwd"):  # the forward's workspace, kept since: its packed q | k | v pieces are reused
            _lib.check(lib.r3d_attention_bwd_ep(_p(qkv), 192, B, N, _p(feat[:, 64:128]), feat.stride(0), _p(dfeat[:, 64:128]),
                                                dfeat.stride(0), _p(lse), p_drop, ctypes.c_uint(seed & 0xffffffff),
                                                _p(ctx.seed_dev), seg.clouds, 1.0 / att.temperature, _p(dqkv), 192, _p(aws), 1,
                                                _st()))
        dWqkv = gemm_tn(dqkv, level2)
        for k, m in enumerate((att.q_map, att.k_map, att.v_map)):
            g[m.weight] = dWqkv[64 * k:64 * (k + 1)].reshape(m.weight.shape)
        Wraw = torch.cat([m.weight.reshape(64, -1) for m in (att.q_map, att.k_map, att.v_map)], 0)
        conv_acc(dqkv, Wraw.t().contiguous(), dlevel2)
        # --- point MLP (dgcnn.py:121-122)
        dcat = torch.zeros(M, cat.shape[1], device=dev, dtype=torch.float32)
        add_cols(dfeat[:, :64], dcat[:, :64])
        d = dlevel2
        for jn in reversed(range(len(mlp_saved))):
            conv, bnm = enc.conv.layer[3 * jn], enc.conv.layer[3 * jn + 1]
            dW, dg, db, _, dX = conv_bn_bwd(mlp_saved[jn], d, want_dx=True, dx_acc=dcat if jn == 0 else None)
            g[conv.weight] = dW.view_as(conv.weight)
            g[bnm.weight], g[bnm.bias] = dg, db
            d = dX
        # --- EdgeConv stack, last layer first (dgcnn.py:115-119)
        for l in reversed(range(enc.n_edgeconv)):
            ec = enc.edge_convs[l]
            dx_acc = dcat[:, 64 * (l - 1):64 * l] if l > 0 else None
            dW1, dg1, db1, dW2, dg2, db2 = edgeconv_train_bwd(ec_saved[l], dcat[:, 64 * l:64 * (l + 1)], B, N, dx_acc)
            g[ec.layer[0].weight], g[ec.layer[1].weight], g[ec.layer[1].bias] = dW1, dg1, db1
            g[ec.layer[3].weight], g[ec.layer[4].weight], g[ec.layer[4].bias] = dW2, dg2, db2
        ctx.saved = None
        return (None, None, None) + tuple(g.get(id(p)) for p in ctx.param_list)


def encoder_params(model):
    ps = list(model.encoder.parameters()) + list(model.base_learner.parameters()) + list(model.att_learner.parameters())
    return ps


def get_features_train(model, x, seed, seg=None):
    """feat (B*N, 192) with gradient edges to the encoder / base / attention parameters.  `seg` (ops.SegLayout with
    E = 1, Q > 0): x holds the support clouds followed by the query clouds of an episode and the two getFeatures
    results are returned (see EncoderTrainFn)."""
    params = encoder_params(model)

    class _Fn(EncoderTrainFn):
        @staticmethod
        def forward(ctx, x, *ps):
            ctx.param_list = params  # the module's own Parameter objects (gradient dict is keyed by identity)
            ctx.seg = seg
            feat = EncoderTrainFn.forward(ctx, x, model, seed, *ps)
            if seg is None or seg.Q == 0:
                return feat
            assert seg.E == 1
            rows = seg.rows_a
            ctx.seg_shapes = ((rows, feat.shape[1]), (feat.shape[0] - rows, feat.shape[1]))
            return feat[:rows], feat[rows:]  # the two getFeatures results

        @staticmethod
        def backward(ctx, *dfeats):
            if len(dfeats) == 2:
                dfeats = [d if d is not None else torch.zeros(sh, device=ctx.saved[3].device, dtype=torch.float32)
                          for d, sh in zip(dfeats, ctx.seg_shapes)]
                dfeat = torch.cat(dfeats, 0)
            else:
                dfeat = dfeats[0]
            out = EncoderTrainFn.backward(ctx, dfeat)
            return (None,) + out[3:]

    return _Fn.apply(x, *params)


def mpti_train_forward(model, support_x, support_y, query_x, query_y, gt_support_y, gt_query_y, logger, support_flag):
    from . import head_train
    return head_train.mpti_train_forward(model, support_x, support_y, query_x, query_y, gt_support_y, gt_query_y, logger,
                                         support_flag)
