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


def test_h5_without_any_hdf5_library_says_so(tmp_path, monkeypatch):
    """No h5py and no HDF5 C library: the error names what is missing (the .h5 path itself: tests/test_episode_h5.py)."""
    from r3dfsseg_amd import h5lite
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py is importable here")
    except ImportError:
        pass
    monkeypatch.setattr(h5lite, "_LIB", None)
    monkeypatch.setattr(h5lite, "_candidates", lambda: iter(()))
    with pytest.raises(h5lite.H5Error, match="HDF5"):
        EIO.read_episode(str(tmp_path / "0.h5"))


@pytest.mark.gpu
def test_collate_on_device_matches_host(tmp_path):
    _, data, raw = _raw_episode(seed=9)
    host, _ = EIO.collate_test(raw, "cpu")
    dev, _ = EIO.collate_test(raw, "cuda")
    for a, b in zip(host, dev):
        assert b.is_cuda and torch.equal(a, b.cpu())


def test_collate_hands_over_views_of_the_point_major_rows():
    """As the reference's collate (loader.py:1679: from_numpy(x).transpose(2, 3)): no copy, no transpose."""
    from r3dfsseg_amd import ops
    _, data, raw = _raw_episode(seed=4)
    out, _ = EIO.collate_test(raw, "cpu")
    for i in (0, 2):
        assert not out[i].is_contiguous() and ops.is_point_major_view(out[i])
        assert torch.equal(out[i], data[i])
    assert not ops.is_point_major_view(data[0].contiguous())
    both = ops.cat_clouds(out[0].reshape(4, 9, 256), out[2], 0)  # support + query clouds: still point-major rows
    assert ops.is_point_major_view(both) and torch.equal(both, torch.cat((data[0].reshape(4, 9, 256), data[2]), 0))


@pytest.mark.gpu
def test_point_major_input_gives_the_same_bits_as_channel_major(tmp_path):
    """The episode as the collate hands it over (views of point-major rows; the encoder reads them as they lie and the
    first kNN packs its own operand) against the same episode as contiguous channel-major tensors: identical logits,
    eval forward and training step."""
    from types import SimpleNamespace
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg, data, raw = _raw_episode(seed=9)
    sd = S.make_state_dict(cfg, 123)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().eval()
    dev, _ = EIO.collate_test(raw, "cuda")
    sy, qy = dev[1], dev[3]
    with torch.no_grad():
        a = m(dev[0], sy, dev[2], qy, lp_iters=m.lp_max_iter)
        b = m(dev[0].contiguous(), sy, dev[2].contiguous(), qy, lp_iters=m.lp_max_iter)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    m.train()
    m.att_learner.dropout.p = 0.0
    tdata, _ = S.make_episode(cfg, seed=9, noise_ratio=0.2, train=True)
    t = [x.cuda() for x in tdata]
    outs = []
    for pm in (True, False):
        sx = t[0].transpose(2, 3).contiguous().transpose(2, 3) if pm else t[0].contiguous()
        qx = t[2].transpose(1, 2).contiguous().transpose(1, 2) if pm else t[2].contiguous()
        m.load_state_dict(sd)
        m.zero_grad()
        out = m(sx, t[1], qx, t[3], gt_support_y=t[6], gt_query_y=t[7], train=True, support_flag=t[10], lp_iters=m.lp_max_iter)
        (out[1] + 0.1 * out[2]).backward()
        outs.append((out[0].detach().clone(), m.encoder.edge_convs[0].layer[0].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
def test_batched_eval_from_collated_views_equals_channel_major():
    """EpisodeBatch keeps point-major views (stacking copies rows, x_all is again a view) and the batched forward gives
    the bits of the same episodes handed over as contiguous channel-major tensors."""
    from types import SimpleNamespace
    from r3dfsseg_amd import ops
    from r3dfsseg_amd.batch import EpisodeBatch
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=256)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(S.make_state_dict(cfg, 123))
    m.cuda().eval()
    views, dense = [], []
    for seed in (3, 4, 5):
        _, data, raw = _raw_episode(seed=seed)
        out, _ = EIO.collate_test(raw, "cuda")
        views.append(out)
        dense.append([t.contiguous() for t in out])
    bv, bd = EpisodeBatch.from_episodes(views), EpisodeBatch.from_episodes(dense)
    assert ops.is_point_major_view(bv.x_all) and bd.x_all.is_contiguous() and torch.equal(bv.x_all, bd.x_all)
    with torch.no_grad():
        lv, _ = m.forward_episodes(bv, lp_iters=m.lp_max_iter)
        ld, _ = m.forward_episodes(bd, lp_iters=m.lp_max_iter)
    assert torch.equal(lv, ld)


@pytest.mark.gpu
def test_feeder_on_the_device_matches_host_collate(tmp_path):
    """The reader thread's copies (its own stream, event handed to the consumer) deliver the files' arrays, in order."""
    _, _, raw = _raw_episode(seed=6)
    want = []
    for i in range(12):
        ep = raw[:3] + (raw[3] + i,) + raw[4:]
        EIO.write_episode(str(tmp_path / ("%d.npz" % i)), ep)
        want.append(EIO.collate_test(ep, "cpu")[0])
    files = EIO.list_episode_files(str(tmp_path))
    busy = torch.zeros(1 << 24, device="cuda")
    got = []
    for out, _ in EIO.EpisodeFeeder(files, device="cuda", depth=4):
        busy.mul_(1.0001)  # kernels in flight on the consumer's stream while the next copies run
        got.append([t.clone() for t in out])
    torch.cuda.synchronize()
    assert len(got) == 12
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert a.is_cuda and torch.equal(a.cpu(), b)
