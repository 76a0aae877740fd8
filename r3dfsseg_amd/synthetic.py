"""Synthetic weights and episodes of the reference's tensor contract.

There is no network for S3DIS / ScanNet blocks or checkpoints, so benchmarks and
parity tests run on synthetic episodes with the layout produced by the reference
collate functions (dataloaders/loader.py:1662-1684) and on random weights with
the reference's state-dict key names (models/mpti.py:60-83, models/dgcnn.py:96-111).
Everything is drawn from ``numpy.random.RandomState`` so that the same seed gives
the same bytes in the build container and on the GPU box.
"""
from collections import OrderedDict

import numpy as np
import torch

DEFAULT_CFG = dict(
    n_way=2, k_shot=5, n_queries=1, pc_in_dim=9, pc_npts=2048, use_attention=True,
    n_subprototypes=100, k_connect=200, sigma=1.0, dgcnn_k=20,
    edgeconv_widths=[[64, 64], [64, 64], [64, 64]], dgcnn_mlp_widths=[512, 256],
    base_widths=[128, 64], output_dim=64, dist_method="cosine", shot_seed=1,
)

# Named workloads of BASELINE.json "configs".
WORKLOADS = {
    "P": dict(n_way=2, k_shot=1, pc_npts=512),          # configs[0] plumbing (ProtoNet)
    "S": dict(n_way=2, k_shot=5, pc_npts=2048),         # configs[1] S3DIS 2-way 5-shot
    "C": dict(n_way=3, k_shot=5, pc_npts=4096),         # configs[3] ScanNet 3-way 5-shot
}


def make_cfg(**over):
    cfg = dict(DEFAULT_CFG)
    cfg.update(over)
    return cfg


def workload_cfg(name, **over):
    cfg = make_cfg(**WORKLOADS[name])
    cfg.update(over)
    return cfg


def _bn(rs, prefix, c, sd, gamma_scale=1.0):
    sd[prefix + ".weight"] = (rs.uniform(0.5, 1.5, c) * gamma_scale).astype(np.float32)
    sd[prefix + ".bias"] = (rs.uniform(-0.2, 0.2, c) * gamma_scale).astype(np.float32)
    sd[prefix + ".running_mean"] = rs.uniform(-0.3, 0.3, c).astype(np.float32)
    sd[prefix + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.array(0, dtype=np.int64)


def _w(rs, shape, fan_in, scale=1.0):
    b = scale / np.sqrt(fan_in)
    return rs.uniform(-b, b, shape).astype(np.float32)


def make_state_dict(cfg, seed=123, feat_scale=0.15):
    """Random weights under the reference's MPTI_SelfAtten state-dict names
    (SURVEY.md 8b).  BN running statistics are non-trivial.  ``feat_scale``
    shrinks the three 64-channel feature groups so that gaussian affinities
    exp(-d^2/2) in the 192-d feature space are neither all ~1 nor all ~0 (a
    trained checkpoint does this by itself)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    cin = cfg["pc_in_dim"]
    n_ec = len(cfg["edgeconv_widths"])
    for i, widths in enumerate(cfg["edgeconv_widths"]):
        in_feat = 2 * cin
        for j, w in enumerate(widths):
            sd["encoder.edge_convs.%d.layer.%d.weight" % (i, 3 * j)] = _w(rs, (w, in_feat, 1, 1), in_feat, 2.0)
            last = (i == 0 and j == len(widths) - 1)
            _bn(rs, "encoder.edge_convs.%d.layer.%d" % (i, 3 * j + 1), w, sd,
                feat_scale if last else 1.0)
            in_feat = w
        cin = widths[-1]
    in_feat = sum(w[-1] for w in cfg["edgeconv_widths"])
    for j, w in enumerate(cfg["dgcnn_mlp_widths"]):
        sd["encoder.conv.layer.%d.weight" % (3 * j)] = _w(rs, (w, in_feat, 1), in_feat, 2.0)
        _bn(rs, "encoder.conv.layer.%d" % (3 * j + 1), w, sd)
        in_feat = w
    d2 = cfg["dgcnn_mlp_widths"][-1]
    in_feat = d2
    nb = len(cfg["base_widths"])
    for j, w in enumerate(cfg["base_widths"]):
        sd["base_learner.convs.%d.0.weight" % j] = _w(rs, (w, in_feat, 1), in_feat, 2.0)
        sd["base_learner.convs.%d.0.bias" % j] = _w(rs, (w,), in_feat)
        _bn(rs, "base_learner.convs.%d.1" % j, w, sd, feat_scale if j == nb - 1 else 1.0)
        in_feat = w
    od = cfg["output_dim"]
    if cfg.get("use_attention", True):
        sd["att_learner.q_map.weight"] = _w(rs, (od, d2, 1), d2, 4.0)
        sd["att_learner.k_map.weight"] = _w(rs, (od, d2, 1), d2, 4.0)
        sd["att_learner.v_map.weight"] = _w(rs, (od, d2, 1), d2, 2.0 * feat_scale)
    else:
        sd["linear_mapper.weight"] = _w(rs, (od, d2, 1), d2, 2.0 * feat_scale)
    feat_dim = cfg["edgeconv_widths"][0][-1] + od + cfg["base_widths"][-1]
    sd["proj.weight"] = _w(rs, (128, feat_dim), feat_dim, 4.0)
    sd["proj.bias"] = _w(rs, (128,), feat_dim)
    assert n_ec >= 1
    return OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items())


def _cloud(rs, N, dup_frac):
    xyz = rs.uniform(0, 1, (N, 3)).astype(np.float32) * np.array([1, 1, 3], np.float32)
    rgb = rs.uniform(0, 1, (N, 3)).astype(np.float32)
    if dup_frac > 0:  # duplicated points (loader.py:171 samples with replacement)
        nd = int(N * dup_frac)
        src = rs.randint(0, N, nd); dst = rs.randint(0, N, nd)
        xyz[dst] = xyz[src]; rgb[dst] = rgb[src]
    xyz = xyz - xyz.min(0)
    XYZ = xyz / xyz.max(0)
    return np.concatenate([xyz, rgb, XYZ], axis=1).astype(np.float32)  # (N, 9)


def _box_mask(rs, pts, lo=0.10, hi=0.40):
    """Points inside a random axis-aligned box covering lo..hi of the cloud."""
    N = pts.shape[0]
    ext = pts[:, :3].max(0)
    for _ in range(64):
        side = rs.uniform(0.4, 0.85, 3).astype(np.float32) * ext
        c0 = rs.uniform(0, 1, 3).astype(np.float32) * (ext - side)
        m = np.all((pts[:, :3] >= c0) & (pts[:, :3] <= c0 + side), axis=1)
        if lo * N <= m.sum() <= hi * N:
            return m
    ax = rs.randint(0, 3)
    m = np.zeros(N, bool)
    m[np.argsort(pts[:, ax])[: max(N // 5, 1)]] = True
    return m


SPLIT_CLASSES = (1, 2, 5, 6, 7, 9)  # six class ids of one cross-validation split, as dataloaders/s3dis.py:30


def make_noise_episode(cfg, seed=0, noise_ratio=0.4, noise_mode="ood", train=False, sampled_classes=None):
    """An episode built by the reference's noise rules (episode_sampler.NoiseEpisodeSampler, loader.py:648-890) on
    synthetic blocks: noise_mode 'ood' draws the noisy shots from classes OUTSIDE the episode, 'sym' from the
    episode's other classes, 'partial' flips a wrong object into the mask (BASELINE.json configs[2] is 'ood' at
    noise_ratio 0.4).  Same return value as make_episode; support_flag holds absolute class ids."""
    from . import episode_sampler as ES
    src = ES.SyntheticBlocks(SPLIT_CLASSES, scans_per_class=4 * cfg["k_shot"] + 8,
                             points_per_block=max(3000, cfg["pc_npts"] * 3 // 2), seed=seed)
    smp = ES.NoiseEpisodeSampler(src, SPLIT_CLASSES, n_way=cfg["n_way"], k_shot=cfg["k_shot"],
                                 n_queries=cfg.get("n_queries", 1), num_point=cfg["pc_npts"],
                                 mode="train" if train else "test", noise_ratio=[noise_ratio] if train else noise_ratio,
                                 noise_type=noise_mode, seed=seed)
    arrays, sc = smp.episode(sampled_classes)
    return (ES.collate_train if train else ES.collate_test)(arrays)


def make_episode(cfg, seed=0, noise_ratio=0.0, dup_frac=0.0, train=False, noise_mode=None):
    """One episode as the reference collate would hand it to a learner
    (channel-major tensors).  Returns (data_list, sampled_classes).
    noise_mode ('ood' | 'sym' | 'partial'): route to make_noise_episode (the reference's own noise rules).

    test layout (loader.py:1680-1682): [support_x, support_y, query_x, query_y,
    support_clusters, query_clusters, gt_support_y]; train layout
    (loader.py:1666-1671) appends gt_query_y, bg_pcd_x, bg_pcd_y, support_flag.
    A noisy shot keeps its (wrong-object) foreground mask in support_y while
    gt_support_y is zeroed for that shot (loader.py:673, 'ood' noise)."""
    if noise_mode is not None:
        return make_noise_episode(cfg, seed, noise_ratio, noise_mode, train)
    rs = np.random.RandomState(seed)
    n_way, k_shot, N = cfg["n_way"], cfg["k_shot"], cfg["pc_npts"]
    n_q = n_way * cfg.get("n_queries", 1)
    sampled_classes = np.arange(1, n_way + 1, dtype=np.int32) * 3
    sx = np.empty((n_way, k_shot, N, 9), np.float32)
    sy = np.zeros((n_way, k_shot, N), np.int32)
    gsy = np.zeros((n_way, k_shot, N), np.int32)
    flag = np.zeros((n_way, k_shot), np.int32)
    n_noise = int(round(k_shot * noise_ratio))
    for w in range(n_way):
        for s in range(k_shot):
            pts = _cloud(rs, N, dup_frac)
            m = _box_mask(rs, pts)
            noisy = s >= k_shot - n_noise
            # class signature in colour so that episodes are learnable / separable
            tint = np.zeros(3, np.float32); tint[(w + (7 if noisy else 0)) % 3] = 0.5
            pts[m, 3:6] = np.clip(pts[m, 3:6] * 0.5 + tint, 0, 1)
            sx[w, s] = pts
            sy[w, s] = m
            gsy[w, s] = 0 if noisy else m
            flag[w, s] = 99 if noisy else sampled_classes[w]
    qx = np.empty((n_q, N, 9), np.float32)
    qy = np.zeros((n_q, N), np.int64)
    for q in range(n_q):
        pts = _cloud(rs, N, dup_frac)
        for w in range(n_way):
            m = _box_mask(rs, pts, 0.08, 0.30) & (qy[q] == 0)
            tint = np.zeros(3, np.float32); tint[w % 3] = 0.5
            pts[m, 3:6] = np.clip(pts[m, 3:6] * 0.5 + tint, 0, 1)
            qy[q][m] = w + 1
        qx[q] = pts
    t = torch.from_numpy
    data = [t(sx).transpose(2, 3).contiguous(), t(sy), t(qx).transpose(1, 2).contiguous(), t(qy),
            t(np.zeros((n_way, k_shot, N), np.int32)), t(np.zeros((n_q, N), np.int32)), t(gsy)]
    if train:
        data += [t(qy.copy()), t(np.zeros((1, 9, N), np.float32)), t(np.zeros((1, N), np.int32)), t(flag)]
    return data, sampled_classes
