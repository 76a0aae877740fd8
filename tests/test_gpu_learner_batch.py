"""MPTILearner_V3.train_batch / test_batch: the reference's learner surface (models/mpti_learner.py:50-102) with E episodes
per call -- what mpti_train_noise.py's loop calls once its DataLoader yields E episodes."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from r3dfsseg_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _learner(cfg, mode="train"):
    from r3dfsseg_amd.mpti_learner import MPTILearner_V3
    L = MPTILearner_V3(SimpleNamespace(**cfg), mode=mode)
    L.model.att_learner.dropout.p = 0.0
    L.model._lp_budget = 150
    return L


def test_train_batch_equals_sequential_episodes_with_one_adam_step():
    """E episodes through train_batch (one launch sequence, one Adam step on the mean gradient) against the same E
    episodes run ONE AT A TIME on the eager path with their gradients accumulated and one Adam step: per episode the
    8-tuple train() returns, the same weights after the step (where the gradient is not rounding noise: Adam turns any
    element's gradient into a step of ~lr), the BatchNorm running statistics bit for bit."""
    from r3dfsseg_amd.dp_train import DPTrainer
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512, pretrain_checkpoint_path="synthetic", model_checkpoint_path=None,
                     lr=1e-3, step_size=5000, gamma=0.5)
    eps = []
    for e in range(4):
        data, _ = S.make_episode(cfg, seed=40 + e, noise_ratio=0.5, train=True)
        eps.append([t.cuda() for t in data])
    A, Bm = _learner(cfg), _learner(cfg)
    # sequential: the reference's own call per episode (model(..., train=True); loss.backward(): mpti_learner.py:60-68),
    # gradients accumulating in the bucket, then ONE optimiser step
    tr = DPTrainer(A)
    tr.step(eps)
    want = tr.last_outputs
    got = Bm.train_batch(eps, None)
    assert len(got) == 4 and all(len(o) == 8 for o in got)
    for e, (o, w) in enumerate(zip(got, want)):
        loss, lp, con, acc = o[:4]
        assert abs(float(loss) - float(w[0])) < 2e-5 and abs(float(lp) - float(w[1])) < 2e-5 and abs(float(con) - float(w[2])) < 2e-5
        wacc = float((w[3].argmax(1) == eps[e][3]).float().mean())
        assert abs(acc - wacc) < 1e-6 and 0.0 <= acc <= 1.0
        np.testing.assert_allclose(np.array([float(v) for v in o[4:]]), w[4].cpu().numpy(), atol=1e-6)
    ga = torch.cat([p.grad.reshape(-1) for p in A.model.parameters()])
    gb = torch.cat([p.grad.reshape(-1) for p in Bm.model.parameters()])
    assert (ga - gb).abs().max().item() <= 2e-6 * ga.abs().max().item()
    pa = torch.cat([p.detach().reshape(-1) for p in A.model.parameters()])
    pb = torch.cat([p.detach().reshape(-1) for p in Bm.model.parameters()])
    solid = ga.abs() >= 1e-3 * ga.abs().max()
    assert (pa - pb).abs()[solid].max().item() < 1e-6 and (pa - pb).abs().max().item() < 2.1e-3
    for (k, va), (_, vb) in zip(A.model.named_buffers(), Bm.model.named_buffers()):
        assert torch.equal(va, vb), k
    assert int(Bm.model.encoder.conv.layer[1].num_batches_tracked.item()) == 8  # two getFeatures calls per episode
    # a second call keeps working (the trainer and its gradient bucket are kept) and the loss goes down
    l0 = float(torch.stack([o[0] for o in got]).mean())
    for _ in range(5):
        got = Bm.train_batch(eps, None)
    assert float(torch.stack([o[0] for o in got]).mean()) < l0


def test_test_batch_equals_test_per_episode():
    cfg = S.make_cfg(n_way=2, k_shot=2, pc_npts=512, pretrain_checkpoint_path=None, model_checkpoint_path="synthetic")
    L = _learner(cfg, mode="test")
    eps = []
    for e in range(3):
        data, sc = S.make_episode(cfg, seed=60 + e, noise_ratio=0.5)
        eps.append([t.cuda() for t in data])
    for ev in (False, True):
        got = L.test_batch(eps, None, eval=ev)
        for e, ep in enumerate(eps):
            pred, loss, acc = L.test(ep, None, eval=ev)
            assert torch.equal(got[e][0], pred)
            assert abs(float(got[e][1]) - float(loss)) < 2e-5 and abs(got[e][2] - acc) < 1e-6
