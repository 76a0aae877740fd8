"""Training-mode launch orchestration (forward with batch-statistics BatchNorm + backward).

Two torch.autograd.Function objects wrap the HIP kernels so that the reference's
``loss.backward(); optimizer.step()`` (models/mpti_learner.py:68-70) works unchanged:
``EncoderTrainFn`` (DGCNN + BaseLearner + SelfAttention, i.e. getFeatures, models/mpti.py:579-589)
and, in head_train.py, the transductive head with its losses.  Nothing here computes: every tensor
operation is a call into libr3d_hip.so; torch provides memory, the autograd graph edge and Adam.
"""
import ctypes
import os

import torch

from . import _lib, ops
from .ops import _p, _rows, _st, _timed

BN_EPS, BN_MOM = 1e-5, 0.1
# False while a training forward must leave the BatchNorm running statistics alone
update_running_stats = True


class BNRecorder:
    """Batch statistics of captured episodes.  Several episodes are in flight on separate streams, so a replay must not
    read-modify-write the shared running statistics; it RECORDS (batch mean, unbiased variance) of every BatchNorm call
    instead -- record 2 e + p for episode e of the step, p = 0 the support pass, 1 the query pass (mpti.py:434,436) --
    and `apply()` folds the records into the running statistics in exactly that order after the step: bit for bit what
    the reference's one-episode-at-a-time schedule gives."""

    def __init__(self, max_episodes, device):
        self.max_records = 2 * max_episodes
        self.device = device
        self.tables = {}        # BatchNorm module -> (records (max_records, 2, C), conv bias or None)
        self.pass_id = 0
        self.index_dev = None   # int32 device word of the replaying slot: 2 e

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


# A/B switch (bench, tests): False runs the support and the query clouds of a training episode as two launch sequences
SHARED_LAUNCHES = os.environ.get("R3D_SHARED_LAUNCHES", "1") != "0"

bn_recorder = None  # set by episode_graph.EpisodeGraphs around the capture of a training episode


def _f(n, dev):
    return torch.empty(n, device=dev, dtype=torch.float32)


# ----------------------------------------------------------------------------- thin wrappers
def colstats(X, C, mode=0, DY=None, bn=None, act=0):
    M, ldx = _rows(X)
    dev = X.device
    lib = _lib.load()
    sums = _f(2 * C, dev)
    ws = _f(lib.r3d_colstats_ws_words(M, C), dev)
    lddy = DY.stride(0) if DY is not None else 0
    sc, sh, mu, is_ = bn if bn is not None else (None, None, None, None)
    with _timed("bn_stats"):
        _lib.check(lib.r3d_colstats(_p(X), ldx, _p(DY), lddy, M, C, mode, _p(sc), _p(sh), _p(mu), _p(is_), act, _p(sums),
                                    _p(ws), _st()))
    return sums


def bn_fold(sums, count, bnmod, bias=None, pass_id=None):
    """Batch mean / invstd -> (scale, shift, mean, invstd); updates the module's running statistics like
    nn.BatchNorm in train mode.  `bias`: conv bias folded away by the mean subtraction (it only shifts the
    running mean).  `pass_id`: which getFeatures call of the episode these statistics belong to (0 support, 1 query)
    when both passes share their launches; None = the recorder's current pass."""
    C = bnmod.num_features
    dev = sums.device
    mean, invstd, scale, shift = _f(C, dev), _f(C, dev), _f(C, dev), _f(C, dev)
    rec = bn_recorder
    upd = update_running_stats and rec is None
    rec_ptr, rec_idx, rec_stride = None, None, 0
    if rec is not None:
        table = rec.slot_for(bnmod, bias)
        rec_ptr, rec_idx, rec_stride = _p(table[rec.pass_id if pass_id is None else pass_id]), _p(rec.index_dev), 2 * C
    _lib.check(_lib.load().r3d_bn_fold(_p(sums), float(count), C, _p(bnmod.weight), _p(bnmod.bias), BN_EPS, BN_MOM,
                                       _p(bnmod.running_mean) if upd else None, _p(bnmod.running_var) if upd else None,
                                       _p(mean), _p(invstd), _p(scale), _p(shift), rec_ptr, rec_idx, rec_stride, _st()))
    if upd:
        if bias is not None:
            bnmod.running_mean.add_(BN_MOM * bias.detach())
        bnmod.num_batches_tracked += 1
    return scale, shift, mean, invstd


def affine_act(Z, scale, shift, act, out=None):
    M, ldz = _rows(Z)
    C = Z.shape[1]
    if out is None:
        out = torch.empty(M, C, device=Z.device, dtype=torch.float32)
    _lib.check(_lib.load().r3d_affine_act(_p(Z), ldz, M, C, _p(scale), _p(shift), act, _p(out), out.stride(0), _st()))
    return out


def bn_bwd_apply(Z, DY, bn, act, sums, count, out=None):
    M, ldz = _rows(Z)
    C = Z.shape[1]
    DZ = torch.empty(M, C, device=Z.device, dtype=torch.float32) if out is None else out
    sc, sh, mu, is_ = bn
    _lib.check(_lib.load().r3d_bn_bwd_apply(_p(Z), ldz, _p(DY), DY.stride(0), M, C, _p(sc), _p(sh), _p(mu), _p(is_), act,
                                            _p(sums), float(count), _p(DZ), DZ.stride(0), _st()))
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
    _lib.check(_lib.load().r3d_pointwise_conv_acc(_p(X), ldx, _p(W), M, X.shape[1], W.shape[0], None, None, 0, _p(out),
                                                  out.stride(0), _st()))


def add_cols(src, dst):
    M, lds = _rows(src)
    _lib.check(_lib.load().r3d_add_cols(_p(src), lds, _p(dst), dst.stride(0), M, src.shape[1], _st()))


# ----------------------------------------------------------------------------- conv + BN + act layer
def _segments(seg, unit=1):
    """[(first row, rows)] of the row segments `seg` (counts in units of `unit` rows); None = one segment."""
    out, r0 = [], 0
    for c in seg:
        out.append((r0 * unit, c * unit))
        r0 += c
    return out


def conv_bn_fwd(X, W2d, bnmod, act, bias=None, out=None, seg_rows=None):
    """Returns (y, saved) with saved = (X, W2d, z, [bn vectors per segment], act, seg_rows).  `seg_rows`: row counts of
    the segments that are normalised separately (support clouds | query clouds); the GEMM runs once over all rows."""
    # raw z (a conv bias cancels under batch statistics) and its column sums from the same GEMM launch
    M, ldx = _rows(X)
    C = W2d.shape[0]
    lib = _lib.load()
    seg_rows = [M] if seg_rows is None else list(seg_rows)
    assert sum(seg_rows) == M and len(seg_rows) in (1, 2)
    z = torch.empty(M, C, device=X.device, dtype=torch.float32)
    sums = _f(2 * C * len(seg_rows), X.device)
    ws = _f(lib.r3d_pointwise_conv_stats_ws_words(M, C), X.device)
    with _timed("pointwise_conv"):
        if len(seg_rows) == 1:
            _lib.check(lib.r3d_pointwise_conv_stats(_p(X), ldx, _p(W2d), M, X.shape[1], C, _p(z), C, _p(sums), _p(ws),
                                                    _st()))
        else:
            _lib.check(lib.r3d_pointwise_conv_stats2(_p(X), ldx, _p(W2d), M, X.shape[1], C, _p(z), C, seg_rows[0],
                                                     _p(sums), _p(sums[2 * C:]), _p(ws), _st()))
    y = out if out is not None else torch.empty(M, C, device=X.device, dtype=torch.float32)
    bns = []
    for s, (r0, rows) in enumerate(_segments(seg_rows)):
        bn = bn_fold(sums[2 * C * s:2 * C * (s + 1)], rows, bnmod, bias, pass_id=s if len(seg_rows) > 1 else None)
        affine_act(z[r0:r0 + rows], bn[0], bn[1], act, out=y[r0:r0 + rows])
        bns.append(bn)
    return y, (X, W2d, z, bns, act, seg_rows)


def conv_bn_bwd(saved, dY, want_dx=True, dx_acc=None):
    """Returns (dW, dgamma, dbeta, dbias, dX), summed over the segments.  dX is accumulated into dx_acc when given."""
    X, W2d, z, bns, act, seg_rows = saved
    C = W2d.shape[0]
    M = z.shape[0]
    dz = torch.empty(M, C, device=z.device, dtype=torch.float32)
    sums = None
    for s, (r0, rows) in enumerate(_segments(seg_rows)):
        zs, dys = z[r0:r0 + rows], dY[r0:r0 + rows]
        sm = colstats(zs, C, mode=1, DY=dys, bn=bns[s], act=act)
        bn_bwd_apply(zs, dys, bns[s], act, sm, rows, out=dz[r0:r0 + rows])
        sums = sm if sums is None else sums + sm
    dW = gemm_tn(dz, X)  # one launch over all rows: the segments' weight gradients add
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
    return dW, sums[C:], sums[:C], dbias, dX


# ----------------------------------------------------------------------------- EdgeConv layer
def edgeconv_train_fwd(inp, idx, ec, B, N, out, seg_clouds=None):
    """`seg_clouds`: cloud counts of the segments whose BatchNorm statistics stay apart (support | query); the PQ GEMM
    runs once over all clouds, the edge passes once per segment."""
    lib = _lib.load()
    dev = inp.device
    W1 = ec.layer[0].weight.reshape(64, -1)
    C = W1.shape[1] // 2
    Wpq = torch.cat((W1[:, :C], W1[:, C:] - W1[:, :C]), 0).contiguous()
    PQ = ops.pointwise_conv(inp, Wpq)
    K = idx.shape[-1]
    seg_clouds = [B] if seg_clouds is None else list(seg_clouds)
    assert sum(seg_clouds) == B
    multi = len(seg_clouds) > 1
    ws = _f(lib.r3d_edgeconv_train_ws_words(), dev)
    W2 = ec.layer[3].weight.reshape(64, 64).contiguous()
    argmax = torch.empty(B * N, 64, device=dev, dtype=torch.int32)
    argmin = torch.empty(B * N, 64, device=dev, dtype=torch.int32)
    zmax = torch.empty(B * N, 64, device=dev, dtype=torch.float32)
    zmin = torch.empty(B * N, 64, device=dev, dtype=torch.float32)
    idx3 = idx.view(B, N, K)
    bn1s, bn2s = [], []
    for s, (b0, Bs) in enumerate(_segments(seg_clouds)):
        r0, r1 = b0 * N, (b0 + Bs) * N
        E = Bs * N * K
        pq, ix = PQ[r0:r1], idx3[b0:b0 + Bs]
        sums1, sums2 = _f(128, dev), _f(128, dev)
        with _timed("edgeconv"):
            _lib.check(lib.r3d_edge_stats1(_p(pq), _p(ix), Bs, N, K, _p(sums1), _p(ws), _st()))
        bn1 = bn_fold(sums1, E, ec.layer[1], pass_id=s if multi else None)
        with _timed("edgeconv"):  # ONE edge-GEMM pass: z2 statistics and per-point max / min of z2
            _lib.check(lib.r3d_edgeconv_train_fwd_minmax(_p(pq), _p(ix), _p(bn1[0]), _p(bn1[1]), _p(W2), Bs, N, K,
                                                         _p(zmax[r0:r1]), _p(zmin[r0:r1]), _p(argmax[r0:r1]),
                                                         _p(argmin[r0:r1]), _p(sums2), _p(ws), _st()))
        bn2 = bn_fold(sums2, E, ec.layer[4], pass_id=s if multi else None)
        with _timed("edgeconv"):  # BN2 + LeakyReLU is monotone per channel: pick max or min, in place
            _lib.check(lib.r3d_edge_select(_p(zmax[r0:r1]), _p(zmin[r0:r1]), _p(argmax[r0:r1]), _p(argmin[r0:r1]),
                                           _p(bn2[0]), _p(bn2[1]), Bs * N, _p(out[r0:r1]), out.stride(0), _st()))
        bn1s.append(bn1)
        bn2s.append(bn2)
    return (inp, idx3, Wpq, PQ, W2, bn1s, bn2s, argmax, zmax, C, seg_clouds)


def edgeconv_train_bwd(saved, dout, B, N, dx_acc):
    """Returns (dW1 (64,2C,1,1), dg1, db1, dW2 (64,64,1,1), dg2, db2); input gradient accumulated into dx_acc."""
    inp, idx3, Wpq, PQ, W2, bn1s, bn2s, argmax, zmax, C, seg_clouds = saved
    lib = _lib.load()
    dev = PQ.device
    K = idx3.shape[-1]
    M = B * N
    dPQ = _f(M * 128, dev).view(M, 128)
    ws = _f(lib.r3d_edgeconv_train_ws_words(), dev)
    # the reverse neighbour list of ALL clouds in one launch: the input gradient is a gather over incoming edges
    # (deterministic, no float atomics); the segments below differ in their BatchNorm statistics only
    rev = torch.empty(lib.r3d_edge_reverse_ws_words(B, N, K), device=dev, dtype=torch.int32)
    with _timed("edgeconv_bwd"):
        _lib.check(lib.r3d_edge_reverse(_p(idx3), B, N, K, _p(rev), rev.numel(), _st()))
    tot = None
    for s, (b0, Bs) in enumerate(_segments(seg_clouds)):
        r0, r1 = b0 * N, (b0 + Bs) * N
        Ms = Bs * N
        bn1, bn2 = bn1s[s], bn2s[s]
        ix, do = idx3[b0:b0 + Bs], dout[r0:r1]
        bn2_sums = colstats(zmax[r0:r1], 64, mode=1, DY=do, bn=bn2, act=ops.ACT_LRELU)
        DY1, BE = _f(Ms * K * 64, dev), _f(Ms * 128, dev)
        dW2, bn1_sums = _f(64 * 64, dev), _f(128, dev)
        with _timed("edgeconv_bwd"):
            _lib.check(lib.r3d_edgeconv_bwd_at(_p(PQ[r0:r1]), _p(ix), _p(bn1[0]), _p(bn1[1]), _p(bn1[2]), _p(bn1[3]), _p(W2),
                                               _p(bn2[0]), _p(bn2[1]), _p(bn2[2]), _p(bn2[3]), _p(bn2_sums), _p(do),
                                               do.stride(0), _p(argmax[r0:r1]), Bs, N, K, _p(DY1), _p(BE), _p(rev), B, b0,
                                               _p(dW2), _p(bn1_sums), _p(dPQ[r0:r1]), _p(ws), _st()))
        part = (dW2, bn1_sums, bn2_sums)
        tot = part if tot is None else tuple(a + b for a, b in zip(tot, part))
    dW2, bn1_sums, bn2_sums = tot
    dWpq = gemm_tn(dPQ, inp)  # (128, C): rows 0..63 = dP^T x, rows 64..127 = dQ^T x
    dW1 = torch.cat((dWpq[:64] - dWpq[64:], dWpq[64:]), 1).reshape(64, 2 * C, 1, 1)
    if dx_acc is not None:
        conv_acc(dPQ, Wpq.t().contiguous(), dx_acc)
    return dW1, bn1_sums[64:], bn1_sums[:64], dW2.view(64, 64, 1, 1), bn2_sums[64:], bn2_sums[:64]


# ----------------------------------------------------------------------------- encoder
class EncoderTrainFn(torch.autograd.Function):
    """getFeatures in training mode.  forward(x (B,C_in,N), model, seed, *params) -> feat (B*N, 192).

    ``ctx.seg_clouds = [S, Q]`` (set by the caller before forward) runs the episode's two getFeatures calls
    (mpti.py:434,436: S support clouds, then Q query clouds) through ONE launch sequence over the S + Q clouds: kNN,
    every GEMM and the attention see all clouds in one grid, every BatchNorm keeps the statistics of the two calls
    apart (and updates / records them in the reference's order: support, then query), and backward returns the SUM of
    the two calls' parameter gradients."""

    @staticmethod
    def forward(ctx, x, model, seed, *params):
        enc, base, att = model.encoder, model.base_learner, model.att_learner
        lib = _lib.load()
        B, _, N = x.shape
        M = B * N
        dev = x.device
        seg_clouds = getattr(ctx, "seg_clouds", None)
        seg_rows = [c * N for c in seg_clouds] if seg_clouds else None
        if seg_clouds:
            assert sum(seg_clouds) == B and seg_rows[0] % 64 == 0, "shared launches need a 64-row aligned support block"
        x = x.contiguous().float()
        x_pm = ops.cm_to_pm(x)
        cat = torch.empty(M, 64 * enc.n_edgeconv, device=dev, dtype=torch.float32)
        inp, ec_saved = x_pm, []
        for l in range(enc.n_edgeconv):
            idx = ops.knn(inp, B, N, enc.k, x_cm=x if l == 0 else None)
            out = cat[:, 64 * l:64 * (l + 1)]
            ec_saved.append(edgeconv_train_fwd(inp, idx, enc.edge_convs[l], B, N, out, seg_clouds))
            inp = out
        h, mlp_saved = cat, []
        for jn in range(len(enc.conv.layer_dims)):
            W = enc.conv.layer[3 * jn].weight
            h, sv = conv_bn_fwd(h, W.reshape(W.shape[0], -1).contiguous(), enc.conv.layer[3 * jn + 1], ops.ACT_LRELU,
                                seg_rows=seg_rows)
            mlp_saved.append(sv)
        level2 = h
        feat = torch.empty(M, model.feat_dim, device=dev, dtype=torch.float32)
        ops.copy_cols(cat[:, :64], feat[:, :64])
        hb, base_saved = level2, []
        for i, seq in enumerate(base.convs):
            last = i == base.num_convs - 1
            W = seq[0].weight
            hb, sv = conv_bn_fwd(hb, W.reshape(W.shape[0], -1).contiguous(), seq[1], ops.ACT_NONE if last else ops.ACT_RELU,
                                 bias=seq[0].bias, out=feat[:, 128:] if last else None, seg_rows=seg_rows)
            base_saved.append(sv)
        Wqkv, qscale = att._fold()
        qkv = ops.pointwise_conv(level2, Wqkv, qscale, None, ops.ACT_NONE)
        lse = torch.empty(M, device=dev, dtype=torch.float32)
        p_drop = float(att.dropout.p)
        aws = _f(lib.r3d_attention_ws_words(B, N), dev)
        with _timed("attention"):
            _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(feat[:, 64:128]), feat.stride(0), _p(lse), p_drop,
                                                   ctypes.c_uint(seed & 0xffffffff), _p(model._slot.seed_dev), _p(aws), _st()))
        ctx.model, ctx.dims, ctx.seed_dev = model, (B, N, seed, p_drop), model._slot.seed_dev
        model._dbg_idx = [sv[1] for sv in ec_saved]  # neighbour lists of this pass (parity tests inject them into the oracle)
        if getattr(model, "_trace", None) is not None:  # one entry per getFeatures call
            for b0, Bs in _segments(seg_clouds or [B]):
                model._trace.setdefault("idx", []).append([sv[1][b0:b0 + Bs] for sv in ec_saved])
                model._trace.setdefault("argmax", []).append([sv[7][b0 * N:(b0 + Bs) * N] for sv in ec_saved])
        ctx.saved = (ec_saved, mlp_saved, base_saved, cat, level2, Wqkv, qkv, lse, feat, aws)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        model = ctx.model
        enc, base, att = model.encoder, model.base_learner, model.att_learner
        B, N, seed, p_drop = ctx.dims
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
            _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(feat[:, 64:128]), feat.stride(0), _p(dfeat[:, 64:128]),
                                                dfeat.stride(0), _p(lse), p_drop, ctypes.c_uint(seed & 0xffffffff),
                                                _p(ctx.seed_dev), 1.0 / att.temperature, _p(dqkv), 192, _p(aws), 1, _st()))
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


def shared_launches_ok(model, n_support_clouds):
    """Support and query clouds may share their launches when the support block ends on a 64-row GEMM tile."""
    return SHARED_LAUNCHES and (n_support_clouds * model.n_points) % 64 == 0


def get_features_train(model, x, seed, seg_clouds=None):
    """feat (B*N, 192) with gradient edges to the encoder / base / attention parameters.  `seg_clouds = [S, Q]`: x holds
    the support clouds followed by the query clouds of an episode (see EncoderTrainFn)."""
    params = encoder_params(model)

    class _Fn(EncoderTrainFn):
        @staticmethod
        def forward(ctx, x, *ps):
            ctx.param_list = params  # the module's own Parameter objects (gradient dict is keyed by identity)
            ctx.seg_clouds = seg_clouds
            feat = EncoderTrainFn.forward(ctx, x, model, seed, *ps)
            if not seg_clouds:
                return feat
            rows = seg_clouds[0] * x.shape[2]
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
