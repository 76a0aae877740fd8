"""Host-side mirror of the reference's models/dgcnn.py + models/attention.py + BaseLearner.

Same class names, constructor arguments and state-dict keys as the reference
(models/dgcnn.py:45-127, models/attention.py:10-48, models/mpti.py:18-40) so reference
checkpoints load unchanged; the torch.nn modules are PARAMETER CONTAINERS only -- the
forward pass runs on the HIP kernels of libr3d_hip.so through r3dfsseg_amd.ops.
"""
import torch
import torch.nn as nn

from . import ops


class conv2d(nn.Module):
    """Parameter container of [Conv2d 1x1 (no bias) -> BN2d -> LeakyReLU(0.2)] x n (dgcnn.py:45-61)."""

    def __init__(self, in_feat, layer_dims, batch_norm=True, relu=True, bias=False):
        super().__init__()
        self.layer_dims = layer_dims
        layers = []
        for i in range(len(layer_dims)):
            in_dim = in_feat if i == 0 else layer_dims[i - 1]
            layers.append(nn.Conv2d(in_dim, layer_dims[i], kernel_size=1, bias=bias))
            if batch_norm:
                layers.append(nn.BatchNorm2d(layer_dims[i]))
            if relu:
                layers.append(nn.LeakyReLU(0.2))
        self.layer = nn.Sequential(*layers)


class conv1d(nn.Module):
    """Parameter container of [Conv1d 1x1 (no bias) -> BN1d -> LeakyReLU(0.2)] x n (dgcnn.py:64-80)."""

    def __init__(self, in_feat, layer_dims, batch_norm=True, relu=True, bias=False):
        super().__init__()
        self.layer_dims = layer_dims
        layers = []
        for i in range(len(layer_dims)):
            in_dim = in_feat if i == 0 else layer_dims[i - 1]
            layers.append(nn.Conv1d(in_dim, layer_dims[i], kernel_size=1, bias=bias))
            if batch_norm:
                layers.append(nn.BatchNorm1d(layer_dims[i]))
            if relu:
                layers.append(nn.LeakyReLU(0.2))
        self.layer = nn.Sequential(*layers)


def _fold_bn(bn, conv_bias=None):
    """Eval-mode BatchNorm as a per-channel affine: y = scale * x + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if conv_bias is not None:
        shift = shift + scale * conv_bias
    return scale.contiguous(), shift.contiguous()


def _refresh(old, new):
    """Write newly folded tensors INTO the storage of the previous cache entry where shapes allow: captured
    episode hipGraphs (episode_graph.py) hold raw pointers to the folded weights, so a weight update must
    refresh them in place instead of replacing (and freeing) them."""
    if isinstance(new, torch.Tensor):
        if (isinstance(old, torch.Tensor) and old.shape == new.shape and old.device == new.device
                and old.dtype == new.dtype):
            if old.data_ptr() != new.data_ptr():
                old.copy_(new)
            return old
        return new
    if isinstance(new, dict):
        return {k: _refresh(old.get(k) if isinstance(old, dict) else None, v) for k, v in new.items()}
    if isinstance(new, (list, tuple)):
        o = old if isinstance(old, (list, tuple)) and len(old) == len(new) else [None] * len(new)
        return type(new)(_refresh(a, b) for a, b in zip(o, new))
    return new


class DGCNN(nn.Module):
    """DGCNN encoder (dgcnn.py:83-127): 3 x {kNN on current features -> EdgeConv -> max over K},
    concat -> point MLP.  forward_pm works on point-major matrices (rows = points)."""

    def __init__(self, edgeconv_widths, mlp_widths, nfeat, k=20, return_edgeconvs=False):
        super().__init__()
        self.n_edgeconv = len(edgeconv_widths)
        self.k = k
        self.return_edgeconvs = return_edgeconvs
        for w in edgeconv_widths:
            if list(w) != [64, 64]:
                raise NotImplementedError("the fused EdgeConv kernel supports edgeconv widths [64, 64] only, got %s" % (w,))
        self.edge_convs = nn.ModuleList()
        for i in range(self.n_edgeconv):
            in_feat = nfeat * 2 if i == 0 else edgeconv_widths[i - 1][-1] * 2
            self.edge_convs.append(conv2d(in_feat, edgeconv_widths[i]))
        in_dim = sum(w[-1] for w in edgeconv_widths)
        self.conv = conv1d(in_dim, mlp_widths)
        self._folded = None
        self.trace = None  # a list while a parity test records the neighbour lists of a pass
        # parity tests only: callable (layer, idx (B, N, k) int32) -> idx that replaces rows of the device's own lists (the
        # reference's choice on its fp32 near-tie rows, tests/test_gpu_golden_head.py); None in every product path
        self.idx_patch = None

    def _fold(self):
        """Fold eval-mode BN into GEMM epilogues (cached until parameters change)."""
        key = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        f = {"ec": [], "mlp": []}
        with torch.no_grad():
            for ec in self.edge_convs:
                W1 = ec.layer[0].weight.reshape(ec.layer[0].weight.shape[0], -1)  # (64, 2C)
                C = W1.shape[1] // 2
                Wa, Wb = W1[:, :C], W1[:, C:]
                s1, t1 = _fold_bn(ec.layer[1])
                Wpq = torch.cat((Wa, Wb - Wa), 0).contiguous()                     # (128, C)
                sc = torch.cat((s1, s1)).contiguous()
                sh = torch.cat((torch.zeros_like(t1), t1)).contiguous()
                W2 = ec.layer[3].weight.reshape(64, 64).contiguous()
                s2, t2 = _fold_bn(ec.layer[4])
                f["ec"].append((Wpq, sc, sh, W2, s2, t2))
            n_mlp = len(self.conv.layer_dims)
            for j in range(n_mlp):
                W = self.conv.layer[3 * j].weight
                W = W.reshape(W.shape[0], -1).contiguous()
                s, t = _fold_bn(self.conv.layer[3 * j + 1])
                f["mlp"].append((W, s, t))
            f = _refresh(self._folded[1] if self._folded is not None else None, f)
        self._folded = (key, f)
        return f

    def forward_pm(self, x_pm, B, N, x_cm=None):
        """x_pm (B*N, C_in) -> (edgeconv concat (B*N, 64*n_edgeconv), level2 (B*N, mlp[-1])).
        x_cm: the same input in the reference's (B, C_in, N) layout, if the caller has it."""
        if self.training:
            raise NotImplementedError("training-mode forward goes through r3dfsseg_amd.train_ops")
        f = self._fold()
        M = B * N
        cat = torch.empty(M, 64 * self.n_edgeconv, device=x_pm.device, dtype=torch.float32)
        inp = x_pm
        for l in range(self.n_edgeconv):
            Wpq, sc, sh, W2, s2, t2 = f["ec"][l]
            idx = ops.knn(inp, B, N, self.k, x_cm=x_cm if l == 0 else None)
            if self.idx_patch is not None:
                idx = self.idx_patch(l, idx.view(B, N, self.k)).view(idx.shape)
            if self.trace is not None:
                self.trace.append(idx)
            PQ = ops.pointwise_conv(inp, Wpq, sc, sh, ops.ACT_NONE)
            out = cat[:, 64 * l:64 * (l + 1)]
            ops.edgeconv(PQ, idx, W2, s2, t2, out, B, N)
            inp = out
        h = cat
        for (W, s, t) in f["mlp"]:
            h = ops.pointwise_conv(h, W, s, t, ops.ACT_LRELU)
        return cat, h

    def forward(self, x):
        """Reference signature: x (B, C, N) -> (edgeconv_0 (B,64,N), out (B,mlp[-1],N))."""
        B, _, N = x.shape
        x_pm, x_cm = ops.input_layouts(x)
        cat, h = self.forward_pm(x_pm, B, N, x_cm=x_cm)
        outs = [ops.pm_to_cm(cat[:, 64 * l:64 * (l + 1)], B, N) for l in range(self.n_edgeconv)]
        out = ops.pm_to_cm(h, B, N)
        if self.return_edgeconvs:
            return outs, out
        return outs[0], out


class SelfAttention(nn.Module):
    """Single-head point self-attention (attention.py:10-48)."""

    def __init__(self, in_channel, out_channel=None, attn_dropout=0.1):
        super().__init__()
        self.in_channel = in_channel
        self.out_channel = out_channel if out_channel is not None else in_channel
        if self.out_channel != 64:
            raise NotImplementedError("the attention kernel is built for out_channel = 64")
        self.temperature = self.out_channel ** 0.5
        self.q_map = nn.Conv1d(in_channel, self.out_channel, 1, bias=False)
        self.k_map = nn.Conv1d(in_channel, self.out_channel, 1, bias=False)
        self.v_map = nn.Conv1d(in_channel, self.out_channel, 1, bias=False)
        self.dropout = nn.Dropout(attn_dropout)
        self._folded = None

    def _fold(self):
        key = tuple(p._version for p in self.parameters())
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        with torch.no_grad():
            W = torch.cat([m.weight.reshape(self.out_channel, -1) for m in (self.q_map, self.k_map, self.v_map)], 0).contiguous()
            scale = torch.ones(3 * self.out_channel, device=W.device)
            scale[: self.out_channel] = 1.0 / self.temperature  # q / sqrt(d), exact for d = 64
            f = _refresh(self._folded[1] if self._folded is not None else None, (W, scale))
        self._folded = (key, f)
        return self._folded[1]

    def forward_pm(self, x_pm, B, N, out, group=0):
        """group > 0: the B clouds are a batch of episodes of `group` clouds each (ops.attention)."""
        if self.training:
            raise NotImplementedError("training-mode forward goes through r3dfsseg_amd.train_ops")
        W, scale = self._fold()
        qkv = ops.pointwise_conv(x_pm, W, scale, None, ops.ACT_NONE)
        ops.attention(qkv, B, N, out, group=group)
        return out

    def forward(self, x):
        B, _, N = x.shape
        out = torch.empty(B * N, 64, device=x.device, dtype=torch.float32)
        self.forward_pm(ops.cm_to_pm(x), B, N, out)
        return ops.pm_to_cm(out, B, N)


class BaseLearner(nn.Module):
    """Conv1d(bias)+BN(+ReLU except last) stack (mpti.py:18-40)."""

    def __init__(self, in_channels, params):
        super().__init__()
        self.num_convs = len(params)
        self.convs = nn.ModuleList()
        for i in range(self.num_convs):
            in_dim = in_channels if i == 0 else params[i - 1]
            self.convs.append(nn.Sequential(nn.Conv1d(in_dim, params[i], 1), nn.BatchNorm1d(params[i])))
        self._folded = None

    def _fold(self):
        key = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        f = []
        with torch.no_grad():
            for seq in self.convs:
                W = seq[0].weight.reshape(seq[0].weight.shape[0], -1).contiguous()
                s, t = _fold_bn(seq[1], seq[0].bias)
                f.append((W, s, t))
            f = _refresh(self._folded[1] if self._folded is not None else None, f)
        self._folded = (key, f)
        return f

    def forward_pm(self, x_pm, out):
        if self.training:
            raise NotImplementedError("training-mode forward goes through r3dfsseg_amd.train_ops")
        f = self._fold()
        h = x_pm
        for i, (W, s, t) in enumerate(f):
            last = i == self.num_convs - 1
            h = ops.pointwise_conv(h, W, s, t, ops.ACT_NONE if last else ops.ACT_RELU, out=out if last else None)
        return h

    def forward(self, x):
        B, _, N = x.shape
        out = torch.empty(B * N, self.convs[-1][0].weight.shape[0], device=x.device, dtype=torch.float32)
        self.forward_pm(ops.cm_to_pm(x), out)
        return ops.pm_to_cm(out, B, N)
