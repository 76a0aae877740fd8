"""Episode cache contract (reference dataloaders/loader.py:1662-1721): container round trip, collate layout."""
import numpy as np
import pytest
import torch

from r3dfsseg_amd import episode_io as EIO, synthetic as S


def _raw_episode(seed=3):
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=256)
    data, classes = S.make_episode(cfg, seed=seed, noise_ratio=0.2)
    sx, sy, qx, qy = data[:4]
    # on-disk clouds are point-major (n_way, k, N, 9) / (n_q, N, 9)
    raw = (sx.transpose(2, 3).contiguous().numpy(), sy.numpy().astype(np.int32), qx.transpose(1, 2).contiguous().numpy(),
           qy.numpy(), np.asarray(classes, dtype=np.int32), np.zeros(sy.shape, np.int32), np.zeros(qy.shape, np.int32),
           data[6].numpy().astype(np.int32) if len(data) > 6 else sy.numpy().astype(np.int32))
    return cfg, data, raw


def test_npz_round_trip_and_collate(tmp_path):
    cfg, data, raw = _raw_episode()
    fn = str(tmp_path / "0.npz")
    EIO.write_episode(fn, raw)
    back = EIO.read_episode(fn)
    for (name, dt), a, b in zip(EIO.EPISODE_FIELDS, raw, back):
        assert b.dtype == dt and np.array_equal(a, b), name
    out, classes = EIO.collate_test(back)
    assert out[0].shape == data[0].shape and torch.equal(out[0], data[0])      # (n_way, k, 9, N) channel-major
    assert out[2].shape == data[2].shape and torch.equal(out[2], data[2])
    assert out[3].dtype == torch.int64 and torch.equal(out[3], data[3])
    assert torch.equal(out[1], data[1].to(torch.int32)) and list(classes) == list(raw[4])
    assert len(out) == 7


def test_feeder_order_and_errors(tmp_path):
    _, _, raw = _raw_episode()
    for i in (2, 0, 10, 1):
        EIO.write_episode(str(tmp_path / ("%d.npz" % i)), raw[:3] + (raw[3] + i,) + raw[4:])
    files = EIO.list_episode_files(str(tmp_path))
    assert [f.split("/")[-1] for f in files] == ["0.npz", "1.npz", "2.npz", "10.npz"]
    got = [int(out[3].min()) - int(raw[3].min()) for out, _ in EIO.EpisodeFeeder(files, device="cpu", depth=2)]
    assert got == [0, 1, 2, 10]
    np.savez(str(tmp_path / "11.npz"), support_ptclouds=raw[0])
    with pytest.raises(RuntimeError):
        list(EIO.EpisodeFeeder([str(tmp_path / "11.npz")], device="cpu"))


def test_h5_needs_h5py(tmp_path):
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            EIO.read_episode(str(tmp_path / "0.h5"))
    else:
        _, _, raw = _raw_episode()
        EIO.write_episode(str(tmp_path / "0.h5"), raw)
        back = EIO.read_episode(str(tmp_path / "0.h5"))
        assert all(np.array_equal(a, b) for a, b in zip(raw, back))


@pytest.mark.gpu
def test_collate_on_device_matches_host(tmp_path):
    _, data, raw = _raw_episode(seed=9)
    host, _ = EIO.collate_test(raw, "cpu")
    dev, _ = EIO.collate_test(raw, "cuda")
    for a, b in zip(host, dev):
        assert b.is_cuda and torch.equal(a, b.cpu())
