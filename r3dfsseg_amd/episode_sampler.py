"""Noise-episode sampler: the rules of the reference's NoiseInMetaTest.generate_one_episode
(dataloaders/loader.py:648-890) and of its block sampler sample_pointcloud_universal (loader.py:202-352), restated
over an abstract source of labelled blocks.

The reference reads S3DIS / ScanNet blocks (`<data_path>/data/<scan>.npy`, rows = xyz rgb label [...] instance) and a
`class2scans` index (dataloaders/s3dis.py:55-79); neither dataset exists in this image, so the same episode
construction runs on `SyntheticBlocks`, a deterministic generator of blocks with per-class objects.  What is kept from
the reference is everything that gives an episode its MEANING:

  * the number of noisy shots per way, int(round(k_shot * noise_ratio))                         loader.py:668-673
  * where a noisy shot's class comes from: 'sym' = another class of THIS episode, 'ood' = a class of the split
    that is NOT in the episode, 'partial' = the way's own class with a wrong object flipped into the mask,
    'pair' = a fixed class pairing, 'train' = any class of the split                            loader.py:675-687,741-756
  * no block is used twice within an episode (black list)                                       loader.py:690-697,760-773
  * a noise class stops being drawn for a way once it holds k_shot - num_noise - 1 of its shots loader.py:800-804
  * a noisy shot keeps the mask of ITS OWN class as support mask, its ground-truth mask is zero  loader.py:777-789,810-816
  * shots are shuffled within a way; support_flag records every shot's ABSOLUTE class id        loader.py:737-739,806,819-834
  * block -> cloud: the target class keeps its share of the points, xyz shifted to the minimum corner, rgb / 255,
    XYZ = xyz / max                                                                              loader.py:219-231,258-276
  * four extra background clouds from classes outside the episode (train layout only)           loader.py:858-886

Host numpy only (as the reference): this is the caller side of the hot path, SURVEY.md 8(f) N3.
"""
import random

import numpy as np


class SyntheticBlocks:
    """Deterministic stand-in for the block files and the class2scans index (dataloaders/s3dis.py:55-79: a block is
    listed under a class when it holds at least 100 points of it).  Block `c<class>_<i>` contains an object of class
    <class> (25-45 % of the points), one or two objects of other classes and unlabeled clutter (class 0).
    Rows: x y z r g b label instance (rgb in 0..255 like the files the reference reads)."""

    def __init__(self, classes, scans_per_class=24, points_per_block=3000, seed=0):
        self.classes = [int(c) for c in classes]
        self.points_per_block = points_per_block
        self.seed = seed
        self.class2scans = {c: ["c%d_%d" % (c, i) for i in range(scans_per_class)] for c in self.classes}

    def load(self, scan_name):
        cls, i = scan_name[1:].split("_")
        cls, i = int(cls), int(i)
        rs = np.random.RandomState((self.seed * 1000003 + cls * 7919 + i) % (2 ** 31))
        n = self.points_per_block + int(rs.randint(-500, 500))
        xyz = rs.uniform(0, 1, (n, 3)) * np.array([1.0, 1.0, 3.0])
        rgb = rs.uniform(0, 255, (n, 3))
        label = np.zeros(n)
        inst = np.zeros(n)
        others = [c for c in self.classes if c != cls]
        rs.shuffle(others)
        objs = [cls] + others[:rs.randint(1, 3)]
        free = np.ones(n, bool)
        for k, c in enumerate(objs):
            frac = rs.uniform(0.25, 0.45) if k == 0 else rs.uniform(0.08, 0.2)
            ax = rs.randint(0, 3)
            order = np.argsort(xyz[:, ax] + rs.uniform(-0.05, 0.05, n))
            order = order[free[order]]
            take = order[: int(frac * n)] if k % 2 == 0 else order[-int(frac * n):]
            free[take] = False
            label[take] = c
            inst[take] = k + 1
            tint = np.zeros(3)
            tint[c % 3] = 128.0
            rgb[take] = np.clip(rgb[take] * 0.5 + tint + 8.0 * (c // 3), 0, 255)  # a class signature in colour
        return np.concatenate([xyz, rgb, label[:, None], inst[:, None]], 1)


def sample_pointcloud(block, num_point, sampled_classes, sampled_class, support, rng, partial_noise=False,
                      pc_attribs="xyzrgbXYZ", pyrng=None):
    """loader.py:202-352 (sample_pointcloud_universal, clean labels): -> (cloud (num_point, 9) f64, label, gt_label).
    support: binary mask of `sampled_class`; query: 1-based position in `sampled_classes`, 0 elsewhere.
    rng: the numpy stream (the reference draws from the global np.random); pyrng: Python's `random` stream, which the
    reference uses for ONE draw, the foreground flip of partial noise (loader.py:325)."""
    sampled_classes = list(sampled_classes)
    N = block.shape[0]
    if partial_noise:  # loader.py:222-223: plain random sample
        inds = rng.choice(np.arange(N), num_point, replace=(N < num_point))
    else:              # loader.py:224-237: the target class keeps its share of the block
        valid = np.nonzero(block[:, 6] == sampled_class)[0]
        n_valid = len(valid) if N < num_point else int(len(valid) / float(N) * num_point)
        inds = np.concatenate([rng.choice(valid, n_valid, replace=False),
                               rng.choice(np.arange(N), num_point - n_valid, replace=(N < num_point))])
    data = block[inds]
    xyz = data[:, 0:3] - np.amin(data[:, 0:3], axis=0)
    parts = []
    if "xyz" in pc_attribs:
        parts.append(xyz)
    if "rgb" in pc_attribs:
        parts.append(data[:, 3:6] / 255.0)
    if "XYZ" in pc_attribs:
        XYZ = xyz - np.amin(xyz, axis=0)
        parts.append(XYZ / np.amax(XYZ, axis=0))
    cloud = np.concatenate(parts, axis=1)
    labels = data[:, 6].astype(np.int64)

    def to_target(lab):
        if support:
            return lab == sampled_class
        out = np.zeros_like(lab)
        for pos, c in enumerate(sampled_classes):
            out[lab == c] = pos + 1
        return out

    target = to_target(labels)
    if partial_noise:  # loader.py:282-300, 341-348: one wrong object joins the mask, sometimes a right one leaves it
        fg_objs = np.unique(data[target.astype(bool)][:, -1])
        objs = list(np.unique(data[:, -1]))
        if len(objs) > 1 and len(np.unique(data[:, 6])) > 1:
            while True:
                obj = rng.choice(objs, 1, replace=False)[0]
                m = data[:, -1] == obj
                if data[m][:, 6][0] != sampled_class:
                    break
            target = target.copy()
            target[m] = True
        if (pyrng.uniform(0, 1) if pyrng is not None else rng.uniform(0, 1)) > 0.7 and len(fg_objs):
            target = target.copy()
            target[data[:, -1] == rng.choice(fg_objs, 1)[0]] = False
    assert np.sum(target) > 0  # loader.py:350
    return cloud, target, to_target(labels)


def _sample_k(source, num_point, scans, sampled_class, sampled_classes, support, rng, partial_noise=False, pyrng=None):
    out = [sample_pointcloud(source.load(s), num_point, sampled_classes, sampled_class, support, rng, partial_noise,
                             pyrng=pyrng)
           for s in scans]
    return (np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out]))


class NoiseEpisodeSampler:
    """generate_one_episode of the reference's NoiseInMetaTest (loader.py:562-890) on a block source.

    mode 'test': noise_ratio is a number, noise_type in {'sym', 'ood', 'partial', 'pair'};
    mode 'train': noise_ratio is a LIST one entry of which is drawn per episode, noise comes from any class of the
    split (loader.py:585-588, 669-671, 686-687)."""

    def __init__(self, source, classes, n_way=2, k_shot=5, n_queries=1, num_point=2048, mode="test", noise_ratio=0.4,
                 noise_type="sym", noise_pair_dict=None, seed=0):
        if mode not in ("train", "test"):
            raise NotImplementedError("Unkown mode %s! [Options: train/test]" % mode)
        if mode == "train":
            noise_type = "train"
            assert isinstance(noise_ratio, list)
        self.source, self.classes = source, np.array(classes)
        self.n_way, self.k_shot, self.n_queries, self.num_point = n_way, k_shot, n_queries, num_point
        self.mode, self.noise_ratio, self.noise_type = mode, noise_ratio, noise_type
        self.noise_pair_dict = noise_pair_dict
        self.rng = np.random.RandomState(seed)    # np.random.seed(seed) in the reference's process
        self.pyrng = random.Random(seed)          # random.seed(seed): loader.py:325 draws from Python's generator

    def sample_classes(self):
        return self.rng.choice(self.classes, self.n_way, replace=False)  # loader.py:618

    def generate_one_episode(self, sampled_classes):
        rng, k_shot = self.rng, self.k_shot
        if self.mode == "train":
            n_noise = int(round(k_shot * rng.choice(self.noise_ratio)))
        else:
            n_noise = int(round(k_shot * self.noise_ratio))
        if self.mode == "train":
            noise_range = list(self.classes)
        elif self.noise_type == "sym":
            noise_range = list(sampled_classes)
        elif self.noise_type == "ood":
            noise_range = [c for c in self.classes if c not in sampled_classes]
        elif self.noise_type in ("partial", "pair"):
            noise_range = None
        else:
            raise NotImplementedError("noise type %s" % self.noise_type)
        black = []
        class2scans = self.source.class2scans

        def unused(cls):
            return [s for s in class2scans[cls] if s not in black]

        sup, mask, gt, flags, qry, qlab, gqlab = [], [], [], [], [], [], []
        for cls in sampled_classes:
            clean = rng.choice(unused(cls), k_shot - n_noise + self.n_queries, replace=False)
            black.extend(clean)
            q_scans, s_scans = clean[:self.n_queries], clean[self.n_queries:]
            s_pc, s_mask, s_gt = _sample_k(self.source, self.num_point, s_scans, cls, sampled_classes, True, rng)
            q_pc, q_lab, q_gt = _sample_k(self.source, self.num_point, q_scans, cls, sampled_classes, False, rng)
            flag = np.zeros(k_shot)
            flag[:len(s_scans)] = cls
            if self.noise_type == "pair":
                if self.noise_pair_dict is None:
                    raise AttributeError("noise_type 'pair' needs noise_pair_dict (commented out in loader.py:592-593)")
                way_range = [self.noise_pair_dict[cls]]
            elif self.noise_type == "partial":
                way_range = [cls]
            else:
                way_range = list(noise_range)
            for i in range(n_noise):
                count = {c: 0 for c in way_range}  # re-created per shot, as loader.py:748 does
                if self.noise_type in ("pair", "partial"):
                    noisy = rng.choice(way_range, 1)[0]
                else:
                    noisy = cls
                    while noisy == cls:
                        noisy = rng.choice(way_range, 1)[0]
                candidates = unused(noisy)
                scan = rng.choice(candidates, 1, replace=False)
                if self.noise_type == "partial":  # loader.py:763-770: a block with fewer than 3 objects or classes is re-drawn
                    def poor(name):
                        blk = self.source.load(name)
                        return len(np.unique(blk[:, -1])) < 3 or len(np.unique(blk[:, 6])) < 3
                    while poor(scan[0]):
                        scan = rng.choice(candidates, 1, replace=False)
                black.extend(scan)
                n_pc, n_mask, n_gt = _sample_k(self.source, self.num_point, scan, noisy, sampled_classes, True, rng,
                                               partial_noise=self.noise_type == "partial", pyrng=self.pyrng)
                s_pc, s_mask, s_gt = (np.concatenate([a, b], 0) for a, b in ((s_pc, n_pc), (s_mask, n_mask), (s_gt, n_gt)))
                count[noisy] += 1
                if count[noisy] == k_shot - n_noise - 1:
                    way_range.remove(noisy)
                flag[len(s_scans) + i] = noisy
            if n_noise > 0:
                s_gt[-n_noise:] = 0
            assert len(s_pc) == k_shot
            order = np.arange(k_shot)
            rng.shuffle(order)
            sup.append(s_pc[order]); mask.append(s_mask[order]); gt.append(s_gt[order]); flags.append(flag[order])
            qry.append(q_pc); qlab.append(q_lab); gqlab.append(q_gt)
        bg_x, bg_y = [], []
        bg_classes = [c for c in self.classes if c not in sampled_classes]
        for _ in range(min(4, len(bg_classes))):  # loader.py:858-886
            c = rng.choice(bg_classes, 1)[0]
            bg_classes.remove(c)
            scan = rng.choice(unused(c), 1, replace=False)
            black.extend(scan)
            pc, m, _ = _sample_k(self.source, self.num_point, scan, c, sampled_classes, True, rng)
            bg_x.append(pc); bg_y.append(m)
        sup, mask, gt = np.stack(sup), np.stack(mask), np.stack(gt)
        self.last_black_list = black
        return (sup, mask, np.concatenate(qry), np.concatenate(qlab), np.zeros(mask.shape, np.int32),
                np.zeros(np.concatenate(qlab).shape, np.int32), gt, np.concatenate(gqlab),
                np.concatenate(bg_x) if bg_x else np.zeros((0, self.num_point, sup.shape[-1])),
                np.concatenate(bg_y) if bg_y else np.zeros((0, self.num_point)), np.stack(flags))

    def episode(self, sampled_classes=None):
        """-> (arrays with the dtypes of NoiseInMetaTest.__getitem__ (loader.py:627-652), sampled_classes).  Train mode:
        12 arrays (support, masks, query, labels, classes, clusters x2, gt masks, gt labels, bg clouds, bg masks,
        support_flag); test mode: the first 8."""
        sc = np.array(sampled_classes) if sampled_classes is not None else self.sample_classes()
        (sup, mask, qry, qlab, sclu, qclu, gt, gqlab, bgx, bgy, flag) = self.generate_one_episode(sc)
        out = [sup.astype(np.float32), mask.astype(np.int32), qry.astype(np.float32), qlab.astype(np.int64),
               sc.astype(np.int32), sclu.astype(np.int32), qclu.astype(np.int32), gt.astype(np.int32)]
        if self.mode == "train":
            out += [gqlab.astype(np.int32), bgx.astype(np.float32), bgy.astype(np.int32), flag.astype(np.int32)]
        self.last_support_flag = flag.astype(np.int32)
        return out, sc


def collate_train(arrays):
    """batch_test_task_collate (loader.py:1662-1672): the 11-tensor train layout, clouds channel-major."""
    import torch
    (sup, mask, qry, qlab, sc, sclu, qclu, gt, gqlab, bgx, bgy, flag) = arrays
    t = torch.from_numpy
    data = [t(sup).transpose(2, 3).contiguous(), t(mask), t(qry).transpose(1, 2).contiguous(), t(qlab), t(sclu), t(qclu),
            t(gt), t(gqlab), t(bgx).transpose(1, 2).contiguous(), t(bgy), t(flag)]
    return data, sc


def collate_test(arrays):
    """batch_test_task_collate_test (loader.py:1676-1683): the 7-tensor test layout."""
    import torch
    (sup, mask, qry, qlab, sc, sclu, qclu, gt) = arrays[:8]
    t = torch.from_numpy
    return [t(sup).transpose(2, 3).contiguous(), t(mask), t(qry).transpose(1, 2).contiguous(), t(qlab), t(sclu), t(qclu),
            t(gt)], sc
