"""The HIP path END TO END against outputs of the REFERENCE's own `MPTI_SelfAtten.forward` / `ProtoNet.forward`
(tests/golden/head_*.npz, protonet.npz; written by oracle/gen_golden_head.py from models/mpti.py:414-577 and
models/protonet.py:245-275 run on torch-CPU -- see that file for the environment it supplies).

The tight statements are split over two links: oracle == reference on these fixtures (tests/test_oracle_golden_head.py,
CPU, 2e-5 / exact indices with the reference's near-tie neighbour rows injected) and HIP == oracle (the chain-of-custody
tests).  Here the device runs end to end, twice per fixture: "free" (nothing injected: the device decides every index on its
own features, so a near-tie may flip a neighbour and move a handful of points) and "patched" (the reference's own choice on
its near-tie rows written over the device's lists: then EVERY logit has to agree).  The fixtures *_S are BASELINE.json
configs[1] / configs[2] at their own size (2-way 5-shot 2048 points; ood noise 0.4 through the clean-shot detection; one
training step).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from r3dfsseg_amd import synthetic as S  # noqa: E402
from test_oracle_golden_head import ALL_FIXTURES, FIXTURES, FIXTURES_S, GOLD, fixture, row_hash  # noqa: E402

pytestmark = pytest.mark.gpu


def _model(cfg, sd, train):
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(sd)
    m.cuda()
    if train:
        m.train()
        m.att_learner.dropout.p = 0.0  # as the generator (the mask is a random draw)
    else:
        m.eval()
    m._trace = {}
    return m


def _reference_near_tie_rows(g, n_support_clouds):
    """DGCNN.idx_patch that writes the REFERENCE's neighbour rows over the device's own wherever the reference's margin
    (k-th against (k+1)-th score of models/dgcnn.py:18-20) is within fp32 rounding or its GEMM order chose differently from
    the channel-ascending chain -- the rows oracle/gen_golden_head.py stores (call = 3 * getFeatures call + layer; the device
    runs the support and the query clouds of an episode as one batch of clouds, support first)."""
    where = g["knnfix_where"].astype(np.int64)
    rows = torch.from_numpy(g["knnfix_idx"].astype(np.int32))

    def patch(layer, idx):
        sel = np.nonzero(where[:, 0] % 3 == layer)[0]
        if not len(sel):
            return idx
        cloud = torch.from_numpy(where[sel, 1] + (where[sel, 0] >= 3) * n_support_clouds).to(idx.device)
        point = torch.from_numpy(where[sel, 2]).to(idx.device)
        idx = idx.clone()
        idx[cloud, point] = rows[sel].to(idx.device)
        return idx
    return patch


def _reference_near_tie_lists(g, seen):
    """MPTI_SelfAtten.nbr_patch: the reference's 201-NN rows (faiss' BLAS formulation, models/mpti.py:733-736) written over
    the device's own on the rows whose margin -- first dropped against last kept squared distance -- is within rounding.
    `seen` receives the number of rows where the device's own SET differed from the reference's, and how many of those lie
    outside the near-tie rows (must be none)."""
    gap, dlast = g["knn_gap"], g["knn_dlast"]
    tie = np.nonzero(np.abs(gap) < 2e-5 * np.maximum(1.0, dlast))[0]
    if "knn_idx" in g.files:
        ref_rows = g["knn_idx"][tie].astype(np.int32)
        want_hash = row_hash(g["knn_idx"].astype(np.int64)[:, 1:])
    else:
        assert np.array_equal(tie, g["knn_tie_rows"])
        ref_rows = g["knn_tie_idx"].astype(np.int32)
        want_hash = g["knn_sethash"]

    def patch(nbr):
        n = len(gap)
        own = nbr[0, :n].cpu().numpy().astype(np.int64)
        bad = np.nonzero(row_hash(own[:, 1:]) != want_hash)[0]
        seen["flipped"], seen["outside"] = len(bad), int((~np.isin(bad, tie)).sum())
        nbr = nbr.clone()
        nbr[0, torch.from_numpy(tie).to(nbr.device)] = torch.from_numpy(ref_rows).to(nbr.device)
        return nbr
    return patch


# Bars = at most 3x what was measured on MI355X (the test prints the figures; round 4).
# patched: the reference's own choice on ITS near-tie rows (encoder kNN and 201-NN) written over the device's lists -- every
#   other row must then agree by itself -- which makes the end-to-end comparison a statement about EVERY point: measured
#   max logit error 5e-6 .. 9e-6 (512-point fixtures), 1.7e-5 .. 1.9e-5 (2048 points x 5 shots), arg-max identical,
#   gradient norms 7e-5 .. 3e-4, sampled gradient entries 4e-4 .. 6e-4 (worst tensor).
# free: nothing injected, the device decides every index on its own features.  A near-tie row may flip; in eval mode that
#   moves a handful of points (measured: 0 .. 2.5 % of the logits beyond 1e-4, arg-max identical).  In TRAINING mode at full
#   size one flipped neighbour changes a max-pooled feature, with it a farthest-point-sampling decision and so the prototype
#   SET (head_train_S: 23 % of the prototypes within 1e-4): the reference run twice with its GEMM's summation order changed
#   would differ from itself the same way, so there the free run only has to reproduce the loss.
def _bars(name, patched, train):
    big = name.endswith("_S")
    if patched:
        b = dict(frac=1.0, frac_p=1.0, emax=6e-5 if big else 3e-5, agree=1.0, dloss=5e-6)
    elif name in ("head_eval", "head_clean") and os.environ.get("R3D_MATRIX_ARITH") != "fp32":
        # (no near-tie row of these two episodes flips in the default arithmetic; with the fp32 matrix kernels one does, and
        # the un-patched run is held to the bars of the other small fixtures)
        b = dict(frac=1.0, frac_p=1.0, emax=3e-5, agree=1.0, dloss=5e-6)
    elif big and train:
        b = dict(frac=0.0, frac_p=0.0, emax=None, agree=0.0, dloss=1e-3)
    elif big:
        b = dict(frac=0.93, frac_p=0.97, emax=None, agree=0.999, dloss=2e-5)
    else:
        b = dict(frac=0.98, frac_p=0.99, emax=None, agree=0.999, dloss=2e-5)
    if train:
        b.update(gnorm=1e-3, gmed=8e-4, gmax=2e-3, grads=patched or not big)
        if os.environ.get("R3D_MATRIX_ARITH") == "fp32":
            # (the fp32 matrix kernels: sampled entries 1.1e-3 median / 2.2e-3 max at head_train_S, norms 2.3e-4 -- their
            # accumulation order inside the MFMA chain differs from the oracle's more than the three-piece form's does)
            b.update(gmed=2e-3, gmax=4e-3)
    return b


@pytest.mark.parametrize("patched", [False, True], ids=["free", "patched"])
@pytest.mark.parametrize("name", list(ALL_FIXTURES))
def test_hip_path_against_reference_outputs(name, patched):
    from r3dfsseg_amd import ops
    cfg, sd, data, mode, g = fixture(name)
    n_way, N = cfg["n_way"], cfg["pc_npts"]
    train = mode == "train"
    bar = _bars(name, patched, train)
    m = _model(cfg, sd, train)
    seen = {}
    if patched:
        m.encoder.idx_patch = _reference_near_tie_rows(g, n_way * cfg["k_shot"])
        m.nbr_patch = _reference_near_tie_lists(g, seen)
    ep = [t.cuda() if torch.is_tensor(t) else t for t in data]
    if train:
        out = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], gt_query_y=ep[7], train=True, support_flag=ep[10],
                lp_iters=m.lp_max_iter)
        (out[1] + 0.1 * out[2]).backward()  # models/mpti_learner.py:66
        assert m.lp_converged(backward=True)
        logits, loss = out[0].detach(), out[1].detach()
    else:
        with torch.no_grad():
            logits, loss = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], eval=(mode == "clean"), lp_iters=m.lp_max_iter)
        assert m.lp_converged()
    hb = m._head[1]
    desc = hb.desc.view(-1, 32)[0].cpu().numpy()

    # clean-shot detection: the shots the reference dropped
    if mode == "clean":
        keep = m._trace["shot_keep"].cpu().view(n_way, cfg["k_shot"]).float().numpy()
        assert np.array_equal(keep, g["clean_flag"])

    # prototypes per segment: the reference's counts (fixture call order: fg way 0.., then bg; device: bg = segment 0),
    # k + 1 = 101 where torch_cluster's float-rounded count says so
    want_m = [int(g[f"nproto{n_way}"])] + [int(g[f"nproto{w}"]) for w in range(n_way)]
    got_m = [int(desc[ops.HD_SEG_M + s]) for s in range(n_way + 1)]
    assert got_m == want_m, (got_m, want_m)
    n_proto = sum(want_m)
    assert int(desc[ops.HD_N_PROTO]) == n_proto
    want_p = np.concatenate([g[f"proto{n_way}"]] + [g[f"proto{w}"] for w in range(n_way)])
    got_p = hb.nodes[:n_proto].cpu().numpy()
    perr = np.abs(got_p - want_p).max(1)
    frac_p = float((perr <= 1e-4).mean())

    # logits / loss
    ref = torch.from_numpy(g["logits"])
    err = (logits.cpu() - ref).abs() / ref.abs().clamp(min=1.0)
    frac = float((err <= 1e-4).float().mean())
    agree = float((logits.cpu().argmax(1) == ref.argmax(1)).float().mean())
    dloss = abs(float(loss) - float(g["loss"]))
    print("%s %s: prototypes within 1e-4: %.4f, logits within 1e-4: %.4f (max %.2e), arg-max agreement %.4f, |loss - ref| %.2e"
          % (name, "patched" if patched else "free", frac_p, frac, float(err.max()), agree, dloss))
    # measured on MI355X, free: prototypes 1.0000, logits 0.993-1.0000 (eval: max 7e-6), arg-max 1.0000, loss 6e-6
    if patched:
        print("%s: 201-NN rows whose set differed from the reference's before the patch: %d (outside its near-tie rows: %d)"
              % (name, seen["flipped"], seen["outside"]))
        assert seen["outside"] == 0
    assert frac_p >= bar["frac_p"] and frac >= bar["frac"] and agree >= bar["agree"] and dloss <= bar["dloss"]
    assert bar["emax"] is None or float(err.max()) <= bar["emax"]

    if train and bar["grads"]:
        assert abs(float(out[2]) - float(g["contrast"])) <= 1e-3 * max(1.0, abs(float(g["contrast"])))
        np.testing.assert_allclose(np.array([float(v) for v in out[3:]]), g["metrics"], atol=3e-3)
        # BatchNorm running statistics after the two getFeatures calls of the step (support, then query)
        sdn = m.state_dict()
        worst = 0.0
        for f in g.files:
            if f.startswith("buf/"):
                got = sdn[f[4:]].detach().cpu().numpy()
                worst = max(worst, float(np.abs(got - g[f]).max()))
                np.testing.assert_allclose(got, g[f], atol=5e-6, rtol=1e-5, err_msg=f)
        # parameter gradients of lp_loss + 0.1 * contrastive: norm and a sample of entries per tensor
        rel = {}
        for pname, p in m.named_parameters():
            if "gnorm/" + pname not in g.files:
                continue
            gn = float(g["gnorm/" + pname])
            gv = p.grad.detach().reshape(-1).cpu().double()
            pick = g["gpick/" + pname]
            e = float(np.linalg.norm(gv.numpy()[pick] - g["gval/" + pname]) / max(np.linalg.norm(g["gval/" + pname]), 1e-12))
            if gn < 1e-6:  # a conv bias in front of a training-mode BatchNorm: the gradient is sum dz = 0 identically; the
                assert float(gv.norm()) < 1e-6, pname  # reference's autograd leaves rounding noise there, the device writes 0
                rel[pname] = (0.0, 0.0)
                continue
            rel[pname] = (abs(float(gv.norm()) - gn) / max(gn, 1e-12), e)
        assert len(rel) == sum(1 for f in g.files if f.startswith("gnorm/")), "every parameter of the reference has a gradient here"
        wn = max(v[0] for v in rel.values())
        ws = sorted(v[1] for v in rel.values())
        print("%s: running statistics max |diff| %.2e; gradient norms max rel diff %.2e; sampled entries rel-L2 median %.2e max %.2e"
              % (name, worst, wn, ws[len(ws) // 2], ws[-1]))
        # measured: statistics 2.4e-7, norms 3e-4, sampled entries 5e-5 .. 3e-4 median, 6e-4 max
        assert wn <= bar["gnorm"] and ws[len(ws) // 2] <= bar["gmed"] and ws[-1] <= bar["gmax"], \
            {k: v for k, v in rel.items() if v[0] > bar["gnorm"] or v[1] > bar["gmed"]}


def test_hip_protonet_against_reference_outputs():
    """BASELINE configs[0] shape on the device path: ProtoNet (models/protonet.py:245-275), both distance methods."""
    from r3dfsseg_amd.protonet import ProtoNet
    cfg = S.make_cfg(n_way=2, k_shot=1, pc_npts=512)
    sd = S.make_state_dict(cfg, seed=123)
    data, _ = S.make_episode(cfg, seed=10)
    g = np.load(os.path.join(GOLD, "protonet.npz"))
    for dm in ("cosine", "euclidean"):
        m = ProtoNet(SimpleNamespace(**dict(cfg, dist_method=dm)))
        m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()})
        m.cuda().eval()
        with torch.no_grad():
            logits, loss = m(data[0].cuda(), data[1].cuda(), data[2].cuda(), data[3].cuda())
        ref = torch.from_numpy(g["logits_" + dm])
        err = (logits.cpu() - ref).abs() / ref.abs().clamp(min=1.0)
        frac = float((err <= 1e-4).float().mean())
        print("protonet %s: logits within 1e-4: %.4f (max %.2e)" % (dm, frac, float(err.max())))
        assert frac >= 0.99 and abs(float(loss) - float(g["loss_" + dm])) <= 1e-4


def test_miou_of_the_device_predictions_against_the_references():
    """north_star: mIoU within +-0.2 pt of the reference.  The query predictions of the reference's own forward (arg-max
    of the stored logits, eval_noise.py:85-95) through the reference's metric (eval_noise.py:23-72, restated in the
    oracle) against the device's predictions through the device histogram, over the two-way eval fixtures."""
    from oracle import r3d_oracle as O
    from r3dfsseg_amd.metrics import MIoUAccumulator
    test_classes = [3, 6, 9, 11]
    acc = MIoUAccumulator(test_classes)
    preds, gts, l2cs = [], [], []
    for i, name in enumerate(("head_eval", "head_clean")):
        cfg, sd, data, mode, g = fixture(name)
        m = _model(cfg, sd, False)
        ep = [t.cuda() if torch.is_tensor(t) else t for t in data]
        with torch.no_grad():
            logits, _ = m(ep[0], ep[1], ep[2], ep[3], gt_support_y=ep[6], eval=(mode == "clean"), lp_iters=m.lp_max_iter)
        l2c = [test_classes[i], test_classes[i + 2]]
        acc.update(logits.argmax(1), ep[3], l2c)
        preds.append(g["logits"].argmax(1)); gts.append(data[3].numpy()); l2cs.append(l2c)
    miou, _ = acc.compute()
    want, _ = O.evaluate_metric(preds, gts, l2cs, test_classes)
    print("mIoU device %.6f reference %.6f" % (miou, want))
    assert abs(miou - want) <= 0.002
