"""N3: `NoiseEpisodeSampler` against episodes produced by the REFERENCE's own sampler (NoiseInMetaTest.__getitem__,
dataloaders/loader.py:613-890, and sample_pointcloud_universal, loader.py:138-352).

tests/golden/sampler_*.npz were written by oracle/gen_golden_sampler.py: the blocks of `SyntheticBlocks` stored in the
reference's on-disk format, the reference's code run on them under a fixed seed, three consecutive episodes per scenario
(test: sym / ood / partial / clean; train with a list of noise ratios).  Equality of every array means that the
restatement draws from the random stream call for call as the reference does.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from r3dfsseg_amd.episode_sampler import NoiseEpisodeSampler, SyntheticBlocks  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# the tables of oracle/gen_golden_sampler.py
SCENARIOS = {
    "sampler_test_sym": ("test", "sym", 0.4, 11),
    "sampler_test_ood": ("test", "ood", 0.4, 12),
    "sampler_test_partial": ("test", "partial", 0.4, 13),
    "sampler_test_clean": ("test", "sym", 0.0, 14),
    "sampler_train": ("train", "sym", [0.0, 0.2, 0.4], 15),
    "sampler_train_b": ("train", "sym", [0.2, 0.4], 16),
}
GEOM = dict(n_way=2, k_shot=5, n_queries=1, num_point=512)
BLOCKS = dict(classes=list(range(12)), scans_per_class=12, points_per_block=1500, seed=3)
EPISODES_PER_SCENARIO = 3
NAMES = ["support_x", "support_y", "query_x", "query_y", "sampled_classes", "support_clusters", "query_clusters",
         "gt_support_y", "gt_query_y", "bg_x", "bg_y", "support_flag"]


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_sampler_reproduces_the_reference_episodes(name):
    mode, noise_type, ratio, seed = SCENARIOS[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sampler = NoiseEpisodeSampler(SyntheticBlocks(**BLOCKS), g["classes"], mode=mode, noise_ratio=ratio,
                                  noise_type=noise_type, seed=seed, **GEOM)
    saw_noise = False
    for e in range(EPISODES_PER_SCENARIO):
        arrays, _ = sampler.episode()
        assert len(arrays) == (12 if mode == "train" else 8)
        for n, a in zip(NAMES, arrays):
            key = "ep%d/%s" % (e, n)
            assert tuple(g[key + "_shape"]) == a.shape, (key, a.shape)
            assert str(g[key + "_dtype"]) == str(a.dtype), (key, a.dtype)
            if n in ("support_x", "query_x", "bg_x"):
                np.testing.assert_allclose(a.astype(np.float64).sum(axis=(-1, -2)), g[key + "_sum"], rtol=1e-12, err_msg=key)
                assert np.array_equal(a.reshape(-1)[g[key + "_pick"]], g[key + "_val"]), key
            elif n in ("support_y", "gt_support_y", "bg_y"):
                assert np.array_equal(np.packbits(a.astype(np.uint8).reshape(-1)), g[key]), key
            else:
                assert np.array_equal(a.astype(np.int64), g[key].astype(np.int64)), key
        sy, gsy = arrays[1], arrays[7]
        saw_noise |= bool((sy.reshape(sy.shape[0], sy.shape[1], -1).any(-1) & ~gsy.reshape(sy.shape[0], sy.shape[1], -1).any(-1)).any())
    if mode == "test" and ratio > 0:
        assert saw_noise  # two of five shots per way carry a mask whose ground truth is zero (loader.py:810-816)
