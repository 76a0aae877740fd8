"""Cached test episodes: the on-disk contract of the reference's episode cache and its collate
(dataloaders/loader.py:1662-1721) feeding the MI355X path.

One episode = eight arrays under fixed names (loader.py:1686-1704):
    support_ptclouds (n_way, k_shot, N, 9) f32   support_masks    (n_way, k_shot, N) i32
    query_ptclouds   (n_q, N, 9) f32             query_labels     (n_q, N) i64
    sampled_classes  (n_way,) i32                support_clusters / query_clusters i32
    gt_support_masks (n_way, k_shot, N) i32
The reference stores them as HDF5 datasets (`<index>.h5`).  This module reads and writes that container -- through
h5py where it is importable, else through the HDF5 C library of the image (h5lite.py; checked against a file the
reference's own write_episode produced under real h5py, tests/golden/episode_ref.h5) -- and the same names in `.npz`.  `collate_test` reproduces batch_test_task_collate_test
(loader.py:1676-1683): clouds are stored POINT-major on disk and the model signature is CHANNEL-major; as in the
reference the collate returns the transposed VIEW, and the encoder consumes the point-major rows directly -- the only
re-layout on the path is the channel-major operand the first kNN packs for itself (ops.input_layouts).
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


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def write_episode(out_filename, data):
    """data: the 8-tuple of loader.py:1686 in EPISODE_FIELDS order.  `.h5`: the reference's container (one contiguous
    dataset per name, loader.py:1690-1701) through h5py where it is importable, else through the HDF5 C library
    (h5lite.py); anything else: `.npz` with the same names."""
    arrays = {name: np.asarray(a, dtype=dt) for (name, dt), a in zip(EPISODE_FIELDS, data)}
    if out_filename.endswith(".h5"):
        h5py = _h5py()
        if h5py is not None:
            with h5py.File(out_filename, "w") as f:
                for name, a in arrays.items():
                    f.create_dataset(name, data=a, dtype=a.dtype)
        else:
            from . import h5lite
            h5lite.write_datasets(out_filename, list(arrays.items()))
    else:
        np.savez(out_filename, **arrays)


def read_episode(file_name):
    """-> the 8-tuple of loader.py:1706-1721 (numpy arrays)."""
    if file_name.endswith(".h5"):
        h5py = _h5py()
        if h5py is not None:
            with h5py.File(file_name, "r") as f:
                got = [f[name][:] for name, _ in EPISODE_FIELDS]
        else:
            from . import h5lite
            try:
                got = h5lite.read_datasets(file_name, [name for name, _ in EPISODE_FIELDS])
            except KeyError as e:
                raise ValueError(str(e)) from e
        return tuple(a.astype(dt, copy=False) for a, (_, dt) in zip(got, EPISODE_FIELDS))
    with np.load(file_name) as f:
        missing = [name for name, _ in EPISODE_FIELDS if name not in f]
        if missing:
            raise ValueError("%s lacks episode datasets %s" % (file_name, missing))
        return tuple(f[name].astype(dt, copy=False) for name, dt in EPISODE_FIELDS)


def _to_device(a, device, pinned):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.device(device).type == "cuda":
        if pinned:  # a copy the device can fetch by itself: asynchronous, and on whatever stream is current
            t = t.pin_memory()
        t = t.to(device, non_blocking=True)
    return t


def _to_channel_major(a, device, pinned=False):
    """(..., N, C) point-major numpy -> the (..., C, N) tensor the model signature asks for, as the reference's collate
    builds it (loader.py:1666,1679: `torch.from_numpy(a).transpose(2, 3)`): a transposed VIEW of the point-major rows,
    after one contiguous host-to-device copy of the raw array.  No transpose runs here or in the model: the encoder takes
    the rows as they lie and its first kNN builds its channel-major operand from them (ops.input_layouts)."""
    return _to_device(a, device, pinned).transpose(-1, -2)


def collate_test(data, device="cpu", pinned=False):
    """batch_test_task_collate_test (loader.py:1676-1683) -> ([support_x, support_y, query_x, query_y,
    support_clusters, query_clusters, gt_support_y], sampled_classes).  pinned: stage through page-locked memory so that
    the copies are asynchronous on the current stream (EpisodeFeeder's copy stream)."""
    sx, sy, qx, qy, classes, sc, qc, gsy = data
    out = [_to_channel_major(sx, device, pinned), _to_device(sy, device, pinned),
           _to_channel_major(qx, device, pinned), _to_device(qy.astype(np.int64), device, pinned),
           _to_device(sc, device, pinned), _to_device(qc, device, pinned), _to_device(gsy, device, pinned)]
    return out, classes


class EpisodeFeeder:
    """Iterate over cached episode files.  A reader thread keeps `depth` episodes ahead of the consumer: it decodes the
    file and -- on a GPU -- copies the arrays to the device itself on a COPY STREAM of its own, so that neither the disk
    read nor the host-to-device copies wait for (or hold up) the kernels the consumer has in flight (a copy issued by the
    consumer on the compute stream queues behind them: the sweep of tools/eval_from_cache.py ran at 617 episodes/s that
    way against 757 resident; page-locking every array first was far worse, 143: eight pinned allocations per episode).
    The consumer's stream waits for the episode's copy event."""

    def __init__(self, file_names, device="cuda", depth=8):
        self.file_names, self.device = list(file_names), device
        self._cuda = torch.device(device).type == "cuda"
        self._copy_stream = torch.cuda.Stream(device=device) if self._cuda else None
        self._q = queue.Queue(maxsize=depth)
        self._t = threading.Thread(target=self._reader, daemon=True)
        self._t.start()

    def _reader(self):
        for fn in self.file_names:
            try:
                data = read_episode(fn)
                if self._cuda:
                    with torch.cuda.stream(self._copy_stream):
                        out = collate_test(data, self.device)  # pageable copies: staged by the runtime, on THIS thread and stream
                        ev = torch.cuda.Event()
                        ev.record(self._copy_stream)
                    self._q.put((fn, (out, ev)))
                else:
                    self._q.put((fn, (collate_test(data, self.device), None)))
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
            (out, classes), ev = data
            if ev is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                for t in out:  # allocated on the copy stream, used on the consumer's
                    t.record_stream(cur)
            yield out, classes

    def __len__(self):
        return len(self.file_names)


def list_episode_files(directory):
    """The cache directory layout of loader.py:1623-1636: `<index>.h5` (or `.npz`) files, in index order."""
    names = [f for f in os.listdir(directory) if f.endswith((".h5", ".npz"))]
    names.sort(key=lambda f: (int(os.path.splitext(f)[0]) if os.path.splitext(f)[0].isdigit() else 1 << 60, f))
    return [os.path.join(directory, f) for f in names]
