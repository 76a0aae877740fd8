"""GPU parity: clean-shot detection (a15), ProtoNet head (a16), mIoU accumulator (N1), learners."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import r3d_oracle as O
from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _model(cls, cfg):
    m = cls(SimpleNamespace(**cfg))
    sd = S.make_state_dict(cfg, 123)
    m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()})
    return m.cuda().eval(), sd


@pytest.mark.parametrize("noise,seed", [(0.4, 1), (0.0, 2), (0.4, 3)])
def test_clean_shot_detection_vs_oracle(noise, seed):
    from r3dfsseg_amd import ops
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=5, pc_npts=512)
    m, sd = _model(MPTI_SelfAtten, cfg)
    data, _ = S.make_episode(cfg, seed, noise_ratio=noise)
    sx, sy = data[0], data[1]
    Sn = 10
    with torch.no_grad():
        feat = m.getFeatures_pm(sx.reshape(Sn, 9, 512).cuda())
    keep, dbg = ops.clean_shot_detect(feat, sx.cuda(), sy.cuda(), 2, 5, 512, want_debug=True)
    # oracle on the SAME features (the device features, copied back)
    sfeat = ops.pm_to_cm(feat, Sn, 512).cpu().view(2, 5, 192, 512)
    pl, clean_flag = O.mean_pl_support_y_multi_scale(sfeat, sy, sx)
    assert torch.equal(keep.cpu().view(2, 5).float(), clean_flag), (keep.cpu().view(2, 5), clean_flag)
    # real-valued intermediate: row sums of the cosine map at scale (1,1,1)
    for way in range(2):
        seeds = []
        for k in range(5):
            fg = sy[way, k] == 1
            s, _, _ = O.grid_sampling(sx[way, k][:, fg].t(), sfeat[way, k][:, fg].t(), 1, 1, 1)
            seeds.append(s)
        sd_ = torch.nn.functional.normalize(torch.cat(seeds, 0), p=2, dim=1)
        cos = (sd_ @ sd_.t() * (1 - torch.eye(len(sd_)))).pow(3).sum(1)
        np.testing.assert_allclose(dbg[way, 0, :len(cos)].cpu().numpy(), cos.numpy(), atol=1e-4, rtol=1e-4)


def test_mpti_forward_eval_true_vs_oracle():
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    m, sd = _model(MPTI_SelfAtten, cfg)
    data, _ = S.make_episode(cfg, 4, noise_ratio=0.5)
    sx, sy, qx, qy = data[:4]
    with torch.no_grad():
        logits, loss = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda(), eval=True)
    (wl, wloss), aux = O.mpti_forward(sd, cfg, sx, sy, qx, qy, eval=True, return_aux=True)
    agree = (logits.cpu().argmax(1) == wl.argmax(1)).float().mean().item()
    assert agree >= 0.99, agree
    assert abs(loss.item() - wloss.item()) <= 5e-3 * max(1.0, abs(wloss.item()))


@pytest.mark.parametrize("method,n_way", [("cosine", 2), ("euclidean", 2), ("cosine", 5)])
def test_protonet_forward_vs_oracle(method, n_way):
    """BASELINE.json configs[0]: 2-way 1-shot 512 pts, ProtoNet (there on PyTorch CPU; here on HIP); and 5 ways (six
    classes: the similarity rows travel as two planes of four columns)."""
    from r3dfsseg_amd.protonet import ProtoNet
    cfg = S.workload_cfg("P", dist_method=method, n_way=n_way)
    m, sd = _model(ProtoNet, cfg)
    data, _ = S.make_episode(cfg, 6)
    sx, sy, qx, qy = data[:4]
    with torch.no_grad():
        logits, loss = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda())
    wl, wloss = O.protonet_forward(sd, cfg, sx, sy, qx, qy, method)
    scale = max(1.0, wl.abs().max().item())
    bad = ((logits.cpu() - wl).abs() > 1e-3 * scale).any(1).float().mean().item()
    assert bad < 0.03, bad  # near-tie kNN points only
    assert (logits.cpu().argmax(1) == wl.argmax(1)).float().mean().item() > 0.99
    assert abs(loss.item() - wloss.item()) < 5e-3 * max(1.0, abs(wloss.item()))


def test_protonet_unknown_method_raises():
    from r3dfsseg_amd.protonet import ProtoNet
    cfg = S.workload_cfg("P", dist_method="gaussian")  # the reference's own default (mpti_train_noise.py:224)
    m, _ = _model(ProtoNet, cfg)
    data, _ = S.make_episode(cfg, 6)
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            m(*[t.cuda() for t in data[:4]])


def test_miou_accumulator_vs_oracle():
    from r3dfsseg_amd.metrics import MIoUAccumulator
    rs = np.random.RandomState(3)
    test_classes = [3, 6, 9, 11]
    acc = MIoUAccumulator(test_classes)
    preds, gts, l2cs = [], [], []
    for _ in range(24):
        l2c = rs.choice(test_classes, 2, replace=False)
        p, g = rs.randint(0, 3, (2, 2048)), rs.randint(0, 3, (2, 2048))
        preds.append(p); gts.append(g); l2cs.append(l2c)
        acc.update(torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda(), l2c)
    miou, iou = acc.compute()
    want_miou, want_iou = O.evaluate_metric(preds, gts, l2cs, test_classes)
    np.testing.assert_allclose(iou, want_iou, rtol=1e-12)
    assert abs(miou - want_miou) < 1e-12


def test_learners_test_path():
    from r3dfsseg_amd.mpti_learner import MPTILearner_V3
    from r3dfsseg_amd.proto_learner import ProtoLearner
    cfg = S.workload_cfg("P", model_checkpoint_path="synthetic", pretrain_checkpoint_path=None)
    data, classes = S.make_episode(cfg, 8)
    data = [t.cuda() for t in data]
    L = MPTILearner_V3(SimpleNamespace(**cfg), mode="test")
    pred, loss, acc = L.test(data, classes, eval=True)
    assert pred.shape == (2, 512) and pred.dtype == torch.int64 and 0.0 <= acc <= 1.0 and torch.isfinite(loss)
    P = ProtoLearner(SimpleNamespace(**cfg), mode="test")
    pred, loss, acc = P.test(data, classes)
    assert pred.shape == (2, 512) and 0.0 <= acc <= 1.0 and torch.isfinite(loss)
    with pytest.raises(ValueError):
        MPTILearner_V3(SimpleNamespace(**cfg), mode="bogus")


def test_miou_of_the_pipeline_within_0p2_pt_of_the_oracle():
    """north_star: mIoU within +-0.2 pt of the reference.  12 noisy 2-way episodes over 4 test classes, clean-shot
    detection on (eval=True, as eval_noise.py:57): predictions of the HIP pipeline and of the CPU oracle feed the
    same metric (eval_noise.py:23-72); the pipeline side goes through the device histogram."""
    from r3dfsseg_amd.metrics import MIoUAccumulator
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512)
    sd = S.make_state_dict(cfg, 123)
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda().eval()
    test_classes = [3, 6, 9, 11]
    acc = MIoUAccumulator(test_classes)
    rs = np.random.RandomState(11)
    preds, gts, l2cs = [], [], []
    for e in range(12):
        data, _ = S.make_episode(cfg, seed=300 + e, noise_ratio=0.4)
        sx, sy, qx, qy = data[:4]
        l2c = rs.choice(test_classes, 2, replace=False)
        with torch.no_grad():
            logits, _ = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda(), eval=True)
            if not m.lp_converged():
                logits, _ = m(sx.cuda(), sy.cuda(), qx.cuda(), qy.cuda(), eval=True, lp_iters=m.lp_max_iter)
        acc.update(logits.argmax(1), qy.cuda(), l2c)
        want_logits, _ = O.mpti_forward(sd, cfg, sx, sy, qx, qy, eval=True)
        preds.append(want_logits.argmax(1).numpy()); gts.append(qy.numpy()); l2cs.append(l2c)
    miou, _ = acc.compute()
    want_miou, _ = O.evaluate_metric(preds, gts, l2cs, test_classes)
    assert abs(miou - want_miou) * 100.0 < 0.2, (miou, want_miou)
