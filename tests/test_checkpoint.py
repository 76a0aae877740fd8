"""N4: reference-format checkpoints (utils/checkpoint_util.py:9-50, mpti_train_noise.py:135-152) round-trip through
the MI355X modules.  File handling runs on the CPU; what the kernels see after a load needs the GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from r3dfsseg_amd import checkpoint_util as CK
from r3dfsseg_amd import synthetic as S


def _args(**over):
    cfg = S.make_cfg(n_way=2, k_shot=1, pc_npts=512, lr=1e-3, step_size=5000, gamma=0.5,
                     pretrain_checkpoint_path=None, model_checkpoint_path=None)
    cfg.update(over)
    return cfg


def _write_reference_files(tmp, cfg, seed=321):
    """A training checkpoint and a pre-training checkpoint exactly as the reference writes them."""
    sd = S.make_state_dict(cfg, seed)
    opt_state = {"state": {}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0,
                                                "amsgrad": False, "params": list(range(30))}]}
    torch.save({"iteration": 2000, "model_state_dict": sd, "optimizer_state_dict": opt_state, "loss": np.float64(0.31),
                "IoU": np.float64(0.4242)}, os.path.join(tmp, "checkpoint.tar"))
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc["segmenter.0.weight"] = torch.zeros(3, 3)  # the pre-training segmentor carries tensors the few-shot model lacks
    pre = os.path.join(tmp, "pretrain.tar")
    torch.save({"params": enc}, pre)
    return sd, pre


def test_reference_file_formats_load_on_cpu(tmp_path):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = _args()
    sd, pre = _write_reference_files(str(tmp_path), cfg)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    ptrs = {k: v.data_ptr() for k, v in m.state_dict(keep_vars=True).items()}
    # pre-training file: encoder tensors only, keys re-prefixed, extras ignored
    CK.load_pretrain_checkpoint(m, pre)
    after = m.state_dict()
    for k in after:
        if k.startswith("encoder."):
            assert torch.equal(after[k], sd[k]), k
        else:
            assert torch.equal(after[k], before[k]), k
    # training file: every tensor, in place (captured graphs hold the storage)
    out = CK.load_model_checkpoint(m, str(tmp_path), mode="test")
    assert out is m and m.checkpoint_meta["iteration"] == 2000 and abs(m.checkpoint_meta["IoU"] - 0.4242) < 1e-12
    for k, v in m.state_dict(keep_vars=True).items():
        assert torch.equal(v.detach(), sd[k]) and v.data_ptr() == ptrs[k], k
    assert m.checkpoint_meta["missing"] == [] and m.checkpoint_meta["unexpected"] == []
    # error behaviour of the reference: ValueError for a missing path / missing pre-training path
    with pytest.raises(ValueError):
        CK.load_model_checkpoint(m, str(tmp_path / "nowhere"), mode="test")
    with pytest.raises(ValueError):
        CK.load_pretrain_checkpoint(m, None)
    # a tensor of the wrong shape never reaches the kernels
    bad = dict(sd)
    bad["att_learner.q_map.weight"] = torch.zeros(64, 128, 1)
    torch.save({"iteration": 1, "model_state_dict": bad, "IoU": 0.0}, os.path.join(str(tmp_path), "checkpoint.tar"))
    with pytest.raises(ValueError, match="att_learner.q_map.weight"):
        CK.load_model_checkpoint(m, str(tmp_path), mode="test")


def test_partial_state_dict_is_non_strict(tmp_path):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = _args()
    sd = S.make_state_dict(cfg, 5)
    part = {k: v for k, v in sd.items() if not k.startswith("proj.")}
    part["some.other.head"] = torch.zeros(2)
    torch.save({"iteration": 7, "model_state_dict": part, "IoU": 0.1}, os.path.join(str(tmp_path), "checkpoint.tar"))
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    CK.load_model_checkpoint(m, str(tmp_path), mode="test")   # strict=False in the reference (checkpoint_util.py:34)
    assert sorted(m.checkpoint_meta["missing"]) == ["proj.bias", "proj.weight"]
    assert m.checkpoint_meta["unexpected"] == ["some.other.head"]


@pytest.mark.gpu
def test_learner_from_checkpoint_matches_direct_load_and_graphs_follow(tmp_path):
    from r3dfsseg_amd.episode_graph import EpisodeGraphs
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    from r3dfsseg_amd.mpti_learner import MPTILearner_V3
    cfg = _args(pc_npts=512, k_shot=2)
    # a trainer writes checkpoint.tar ...
    Ltrain = MPTILearner_V3(SimpleNamespace(**dict(cfg, pretrain_checkpoint_path="synthetic")), mode="train")
    ep, _ = S.make_episode(cfg, 5, noise_ratio=0.5, train=True)
    ep = [t.cuda() for t in ep]
    for _ in range(2):
        Ltrain.train(ep, None)
    CK.save_model_checkpoint(Ltrain, str(tmp_path), iteration=2, loss=0.5, iou=0.25)
    CK.save_pretrain_checkpoint(Ltrain.model, str(tmp_path), epoch=3)
    sd = {k: v.detach().cpu().clone() for k, v in Ltrain.model.state_dict().items()}
    # ... a tester reads it (eval_noise.py:125 -> mpti_learner.py:44-46)
    Ltest = MPTILearner_V3(SimpleNamespace(**dict(cfg, model_checkpoint_path=str(tmp_path))), mode="test")
    direct = MPTI_SelfAtten(SimpleNamespace(**cfg))
    direct.load_state_dict(sd)
    direct.cuda().eval()
    data, classes = S.make_episode(cfg, 9)
    data = [t.cuda() for t in data]
    pred, loss, acc = Ltest.test(data, classes)
    with torch.no_grad():
        want, wloss = direct(*data[:4], lp_iters=direct.lp_max_iter)
        got, _ = Ltest.model(*data[:4], lp_iters=direct.lp_max_iter)
    assert torch.equal(got, want) and torch.equal(pred, want.argmax(1))
    # ... and a trainer resumes from it: weights and Adam moments (mpti_learner.py:40-43)
    Lres = MPTILearner_V3(SimpleNamespace(**dict(cfg, model_checkpoint_path=str(tmp_path))), mode="train")
    a, b = Ltrain.optimizer.state_dict(), Lres.optimizer.state_dict()
    assert len(b["state"]) == len(a["state"]) > 0
    for k in a["state"]:
        assert torch.equal(a["state"][k]["exp_avg"].cpu(), b["state"][k]["exp_avg"].cpu())
        assert torch.equal(a["state"][k]["exp_avg_sq"].cpu(), b["state"][k]["exp_avg_sq"].cpu())
    # the pre-training format initialises a fresh trainer's encoder only
    Lpre = MPTILearner_V3(SimpleNamespace(**dict(cfg, pretrain_checkpoint_path=os.path.join(str(tmp_path), "checkpoint_3.tar"))),
                          mode="train")
    for k, v in Lpre.model.state_dict().items():
        if k.startswith("encoder."):
            assert torch.equal(v.cpu(), sd[k]), k
    # captured episode graphs hold raw pointers to BatchNorm-folded copies: a load must reach them
    other = MPTI_SelfAtten(SimpleNamespace(**cfg))
    other.load_state_dict(S.make_state_dict(cfg, 999))
    other.cuda().eval()
    g = EpisodeGraphs(other, data[:4], n_slots=2, train=False)
    out = torch.empty(1, *want.shape, device="cuda")
    g.run([data[:4]], logits_out=out)
    torch.cuda.synchronize()
    assert (out[0] - want).abs().max().item() > 1e-3          # other weights, other logits
    CK.load_model_checkpoint(other, str(tmp_path), mode="test")
    g.run([data[:4]], logits_out=out)
    torch.cuda.synchronize()
    assert g.check()[0] == 0
    np.testing.assert_allclose(out[0].cpu().numpy(), want.cpu().numpy(), atol=2e-5, rtol=1e-5)


class _Evil:
    def __reduce__(self):  # what a crafted checkpoint.tar would carry: a call executed by the unpickler
        return (os.getenv, ("HOME",))


def test_crafted_pickle_is_refused(tmp_path, monkeypatch):
    """A checkpoint that needs more than tensors / containers / numpy scalars must not reach the full unpickler
    unless the caller opts in (pre-trained files of this model come from third parties)."""
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = _args()
    sd = S.make_state_dict(cfg, 5)
    torch.save({"iteration": 1, "model_state_dict": sd, "IoU": 0.0, "loss": _Evil()},
               os.path.join(str(tmp_path), "checkpoint.tar"))
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    monkeypatch.delenv("R3D_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(ValueError):   # the reference's error type for an unusable checkpoint path
        CK.load_model_checkpoint(m, str(tmp_path), mode="test")
    with pytest.raises(RuntimeError, match="R3D_TRUST_CHECKPOINTS"):
        CK._read(os.path.join(str(tmp_path), "checkpoint.tar"))
    monkeypatch.setenv("R3D_TRUST_CHECKPOINTS", "1")  # explicit opt-in: loads (and says so)
    CK.load_model_checkpoint(m, str(tmp_path), mode="test")
    # numpy scalars written by an older numpy (reconstructor under numpy.core) stay loadable without the opt-in
    monkeypatch.delenv("R3D_TRUST_CHECKPOINTS")
    torch.save({"iteration": 3, "model_state_dict": sd, "IoU": np.float32(0.25), "loss": np.float64(1.5)},
               os.path.join(str(tmp_path), "checkpoint.tar"))
    CK.load_model_checkpoint(m, str(tmp_path), mode="test")
    assert m.checkpoint_meta["iteration"] == 3 and abs(m.checkpoint_meta["IoU"] - 0.25) < 1e-7
