"""Generates tests/golden/sampler_*.npz by RUNNING the reference's own noise-episode sampler (build container only).

    python oracle/gen_golden_sampler.py          # needs /root/reference

TEST INFRASTRUCTURE.  `dataloaders/loader.py` (NoiseInMetaTest.__getitem__ / generate_one_episode, loader.py:613-890, and
sample_pointcloud_universal, loader.py:138-352) reads S3DIS blocks from disk; the dataset is not in this image.  This
script writes the blocks of `r3dfsseg_amd.episode_sampler.SyntheticBlocks` into a scratch directory IN THE REFERENCE'S
OWN FORMAT (`<root>/meta/s3dis_classnames.txt`, `<root>/blocks/data/<scan>.npy` with rows x y z r g b label instance,
`<root>/blocks/class2scans.pkl`: dataloaders/s3dis.py:25,47-52) and runs the reference's sampler on them under a fixed
`np.random.seed` / `random.seed`.  Environment supplied, nothing of the reference's logic: `h5py`, `transforms3d`,
`open3d` are imported by loader.py but reached by none of these calls (empty stand-in modules); `np.int` (removed from
numpy 1.24; the reference is older) is `int`.

The episodes come out of the reference's code; tests/test_sampler_golden.py holds `NoiseEpisodeSampler` to them array by
array (it must consume the random stream draw for draw to get there).
"""
import os
import pickle
import random
import shutil
import sys
import tempfile
import types
import io
import contextlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from r3dfsseg_amd.episode_sampler import SyntheticBlocks  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# dataloaders/s3dis.py:21-22 (the meta file itself is not part of the repository)
S3DIS_NAMES = ["ceiling", "floor", "wall", "beam", "column", "window", "door", "table", "chair", "sofa", "bookcase",
               "board", "clutter"]

# name: (mode, noise_type, noise_ratio, seed) -- shared with tests/test_sampler_golden.py by name
SCENARIOS = {
    "sampler_test_sym": ("test", "sym", 0.4, 11),
    "sampler_test_ood": ("test", "ood", 0.4, 12),
    "sampler_test_partial": ("test", "partial", 0.4, 13),
    "sampler_test_clean": ("test", "sym", 0.0, 14),
    "sampler_train": ("train", "sym", [0.0, 0.2, 0.4], 15),
    "sampler_train_b": ("train", "sym", [0.2, 0.4], 16),
}
GEOM = dict(n_way=2, k_shot=5, n_queries=1, num_point=512, cvfold=0)
BLOCKS = dict(classes=list(range(12)), scans_per_class=12, points_per_block=1500, seed=3)
EPISODES_PER_SCENARIO = 3  # consecutive __getitem__ calls on one random stream


def install_environment():
    for name in ("h5py", "transforms3d", "open3d"):
        sys.modules[name] = types.ModuleType(name)
    if not hasattr(np, "int"):
        np.int = int


def write_blocks(root):
    os.makedirs(os.path.join(root, "meta"))
    with open(os.path.join(root, "meta", "s3dis_classnames.txt"), "w") as f:
        f.write("\n".join(S3DIS_NAMES) + "\n")
    data_path = os.path.join(root, "blocks")
    os.makedirs(os.path.join(data_path, "data"))
    src = SyntheticBlocks(**BLOCKS)
    for scans in src.class2scans.values():
        for s in scans:
            np.save(os.path.join(data_path, "data", s + ".npy"), src.load(s))
    c2s = {k: list(src.class2scans.get(k, [])) for k in range(13)}
    with open(os.path.join(data_path, "class2scans.pkl"), "wb") as f:
        pickle.dump(c2s, f)
    return data_path


def digest(arrays, mode):
    """What the fixture keeps of one episode: every integer array whole (bit-packed masks), the clouds as per-cloud
    float64 sums + a sample of entries."""
    names = ["support_x", "support_y", "query_x", "query_y", "sampled_classes", "support_clusters", "query_clusters",
             "gt_support_y"] + (["gt_query_y", "bg_x", "bg_y", "support_flag"] if mode == "train" else [])
    out = {}
    rs = np.random.RandomState(5)
    for n, a in zip(names, arrays):
        a = np.asarray(a)
        out[n + "_shape"] = np.array(a.shape, np.int32)
        out[n + "_dtype"] = np.array(str(a.dtype))
        if n in ("support_x", "query_x", "bg_x"):
            out[n + "_sum"] = a.astype(np.float64).sum(axis=(-1, -2))
            flat = a.reshape(-1)
            pick = rs.randint(0, flat.size, 512)
            out[n + "_pick"] = pick.astype(np.int64)
            out[n + "_val"] = flat[pick]
        elif n in ("support_y", "gt_support_y", "bg_y"):
            assert set(np.unique(a)) <= {0, 1}
            out[n] = np.packbits(a.astype(np.uint8).reshape(-1))
        else:
            out[n] = a.astype(np.int16 if a.size and np.abs(a).max() < 32768 else np.int64)
    return out


def main():
    install_environment()
    from dataloaders.loader import NoiseInMetaTest  # the reference
    root = tempfile.mkdtemp(prefix="r3d_blocks_")
    try:
        data_path = write_blocks(root)
        os.makedirs(OUT, exist_ok=True)
        for name, (mode, noise_type, ratio, seed) in SCENARIOS.items():
            np.random.seed(seed)
            random.seed(seed)
            with contextlib.redirect_stdout(io.StringIO()):
                ds = NoiseInMetaTest(data_path, "s3dis", cvfold=GEOM["cvfold"], num_episode=10, n_way=GEOM["n_way"],
                                     k_shot=GEOM["k_shot"], n_queries=GEOM["n_queries"], mode=mode,
                                     num_point=GEOM["num_point"], pc_attribs="xyzrgbXYZ", pc_augm=False,
                                     noise_ratio=ratio, noise_type=noise_type)
                rec = {"classes": np.array(ds.classes, np.int32)}
                for e in range(EPISODES_PER_SCENARIO):
                    arrays = ds[e]
                    for k, v in digest(arrays, mode).items():
                        rec["ep%d/%s" % (e, k)] = v
            path = os.path.join(OUT, name + ".npz")
            np.savez_compressed(path, **rec)
            print(name, "classes", rec["classes"].tolist(), "episode 0 classes", rec["ep0/sampled_classes"].tolist(),
                  "flags", rec.get("ep0/support_flag", np.zeros(0)).tolist(), os.path.getsize(path))
    finally:
        shutil.rmtree(root)


if __name__ == "__main__":
    main()
