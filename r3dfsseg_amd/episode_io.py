"""Cached test episodes: the on-disk contract of the reference's episode cache and its collate
(dataloaders/loader.py:1662-1721) feeding the MI355X path.

One episode = eight arrays under fixed names (loader.py:1686-1704):
    support_ptclouds (n_way, k_shot, N, 9) f32   support_masks    (n_way, k_shot, N) i32
    query_ptclouds   (n_q, N, 9) f32             query_labels     (n_q, N) i64
    sampled_classes  (n_way,) i32                support_clusters / query_clusters i32
    gt_support_masks (n_way, k_shot, N) i32
The reference stores them as HDF5 datasets (`<index>.h5`).  This module reads and writes the same names in
`.npz` containers natively and in `.h5` files when h5py is importable (it is not part of this image, so the
h5 branch is exercised only where h5py exists).  `collate_test` reproduces batch_test_task_collate_test
(loader.py:1676-1683): clouds are stored POINT-major on disk and the model takes them CHANNEL-major, so the
transpose runs on the GPU (r3d_pm_to_cm) after one contiguous host-to-device copy of the raw array instead
of a strided host transpose + copy.
"""
import os
import queue
import threading

import numpy as np
import torch

EPISODE_FIELDS = (
    ("support_ptclouds", np.float32), ("support_masks", np.int32), ("query_ptclouds", np.float32),
    ("query_labels", np.int64), ("sampled_classes", np.int32), ("support_clusters", np.int32),
    ("query_clusters", np.int32), ("gt_support_masks", np.int32),
)


def _h5():
    try:
        import h5py
    except ImportError as e:  # the image has no h5py: say so instead of guessing at the file format
        raise RuntimeError("reading / writing .h5 episode files needs h5py (not installed here); "
                           "use the .npz container with the same dataset names") from e
    return h5py


def write_episode(out_filename, data):
    """data: the 8-tuple of loader.py:1686 in EPISODE_FIELDS order."""
    arrays = {name: np.asarray(a, dtype=dt) for (name, dt), a in zip(EPISODE_FIELDS, data)}
    if out_filename.endswith(".h5"):
        with _h5().File(out_filename, "w") as f:
            for name, a in arrays.items():
                f.create_dataset(name, data=a, dtype=a.dtype)
    else:
        np.savez(out_filename, **arrays)


def read_episode(file_name):
    """-> the 8-tuple of loader.py:1706-1721 (numpy arrays)."""
    if file_name.endswith(".h5"):
        with _h5().File(file_name, "r") as f:
            return tuple(f[name][:] for name, _ in EPISODE_FIELDS)
    with np.load(file_name) as f:
        missing = [name for name, _ in EPISODE_FIELDS if name not in f]
        if missing:
            raise ValueError("%s lacks episode datasets %s" % (file_name, missing))
        return tuple(f[name].astype(dt, copy=False) for name, dt in EPISODE_FIELDS)


def _to_channel_major(a, device):
    """(..., N, C) point-major numpy -> (..., C, N) tensor on `device`."""
    lead, (N, C) = a.shape[:-2], a.shape[-2:]
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type != "cuda":
        return t.transpose(-1, -2).contiguous()
    from . import ops
    pm = t.reshape(-1, C).to(device, non_blocking=True)           # one contiguous copy, rows = points
    B = int(np.prod(lead)) if lead else 1
    return ops.pm_to_cm(pm, B, N).reshape(*lead, C, N)


def collate_test(data, device="cpu"):
    """batch_test_task_collate_test (loader.py:1676-1683) -> ([support_x, support_y, query_x, query_y,
    support_clusters, query_clusters, gt_support_y], sampled_classes)."""
    sx, sy, qx, qy, classes, sc, qc, gsy = data
    out = [_to_channel_major(sx, device), torch.from_numpy(sy).to(device),
           _to_channel_major(qx, device), torch.from_numpy(qy.astype(np.int64)).to(device),
           torch.from_numpy(sc).to(device), torch.from_numpy(qc).to(device), torch.from_numpy(gsy).to(device)]
    return out, classes


class EpisodeFeeder:
    """Iterate over cached episode files: a reader thread keeps `depth` episodes decoded ahead of the consumer,
    which collates them onto `device` (for episode_graph.EpisodeGraphs.run, learner.test, ...)."""

    def __init__(self, file_names, device="cuda", depth=8):
        self.file_names, self.device = list(file_names), device
        self._q = queue.Queue(maxsize=depth)
        self._t = threading.Thread(target=self._reader, daemon=True)
        self._t.start()

    def _reader(self):
        for fn in self.file_names:
            try:
                self._q.put((fn, read_episode(fn)))
            except Exception as e:  # surface the error in the consumer thread
                self._q.put((fn, e))
        self._q.put(None)

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            fn, data = item
            if isinstance(data, Exception):
                raise RuntimeError("cannot read episode %s" % fn) from data
            yield collate_test(data, self.device)

    def __len__(self):
        return len(self.file_names)


def list_episode_files(directory):
    """The cache directory layout of loader.py:1623-1636: `<index>.h5` (or `.npz`) files, in index order."""
    names = [f for f in os.listdir(directory) if f.endswith((".h5", ".npz"))]
    names.sort(key=lambda f: (int(os.path.splitext(f)[0]) if os.path.splitext(f)[0].isdigit() else 1 << 60, f))
    return [os.path.join(directory, f) for f in names]
