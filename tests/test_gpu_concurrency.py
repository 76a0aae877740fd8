"""Episodes run several at a time on separate HIP streams (episode_graph.py), so kernels of different episodes share
CUs and SIMDs.  No result may depend on what runs beside it: label-propagation solves on one stream while the bf16 x 3
attention kernels (bf16-MFMA-dense) or the fp32 ones keep another stream busy must equal the solve alone bit for bit.
(This failed in ~20 % of the solves while the library still contained packed fp32 VALU arithmetic -- build.py.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_label_propagation_is_not_disturbed_by_attention_on_another_stream():
    from r3dfsseg_amd import _lib
    from r3dfsseg_amd.ops import _p
    lib = _lib.load()
    n, kp1, D = 4396, 201, 192
    torch.manual_seed(0)
    cent = torch.randn(3, D, device="cuda") * 0.5
    x = (cent[torch.randint(0, 3, (n,), device="cuda")] + torch.randn(n, D, device="cuda") * 0.12).contiguous()
    Y = torch.zeros(n, 4, device="cuda")
    Y[torch.arange(300), torch.randint(0, 3, (300,))] = 1
    nd = torch.tensor([n], device="cuda", dtype=torch.int32)
    npd = torch.tensor([300], device="cuda", dtype=torch.int32)
    norm = torch.empty(lib.r3d_knn_norm_ws_words(1, n), device="cuda")
    cm = torch.empty(D * lib.r3d_cm_pitch(n), device="cuda")
    nbr = torch.empty(n, kp1, device="cuda", dtype=torch.int32)
    st = torch.zeros(1, device="cuda", dtype=torch.int32)
    _lib.check(lib.r3d_knn_topk(_p(x), D, None, 1, n, D, kp1, 1, _p(nd), _p(norm), _p(cm), _p(nbr), None, _p(st), None))
    ws = torch.empty(lib.r3d_lp_ws_words(n, kp1), device="cuda", dtype=torch.int32)
    B, N = 12, 2048
    qkv = torch.randn(B * N, 192, device="cuda")
    dO = torch.randn(B * N, 64, device="cuda")
    aws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
    out = torch.empty(B * N, 64, device="cuda")
    lse = torch.empty(B * N, device="cuda")
    dqkv = torch.empty(B * N, 192, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def solves(k):
        res = []
        with torch.cuda.stream(sa):
            for _ in range(k):
                Z = torch.empty(n, 4, device="cuda")
                stats = torch.zeros(2, device="cuda", dtype=torch.int32)
                _lib.check(lib.r3d_label_propagate(_p(x), D, D, _p(nbr), kp1, _p(Y), _p(nd), _p(npd), n, 1.0, 0.99, 200, 1e-6,
                                                   _p(Z), _p(ws), ws.numel(), _p(stats), sa.cuda_stream))
                res.append((Z, stats))
        return res

    def attention(k):
        with torch.cuda.stream(sb):
            for _ in range(k):
                _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.1, 7, None, _p(aws),
                                                       sb.cuda_stream))
                _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.1, 7, None, 0.125,
                                                    _p(dqkv), 192, _p(aws), 1, sb.cuda_stream))

    ref = solves(2)
    torch.cuda.synchronize()
    Zref, sref = ref[0][0].clone(), ref[0][1].tolist()
    assert sref[0] == 1 and torch.equal(ref[1][0], Zref)
    before = lib.r3d_get_matrix_arith()
    try:
        for mode in (1, 0):
            _lib.check(lib.r3d_set_matrix_arith(mode))
            differ = 0
            for rep in range(3):
                attention(40)
                got = solves(50)
                torch.cuda.synchronize()
                differ += sum(1 for Z, s in got if s.tolist() != sref or not torch.equal(Z, Zref))
            assert differ == 0, (mode, differ)
    finally:
        _lib.check(lib.r3d_set_matrix_arith(before))


def test_attention_results_do_not_depend_on_other_streams():
    from r3dfsseg_amd import _lib
    from r3dfsseg_amd.ops import _p
    lib = _lib.load()
    B, N, S = 12, 2048, 4
    torch.manual_seed(3)
    qkv = torch.randn(B * N, 192, device="cuda")
    dO = torch.randn(B * N, 64, device="cuda")

    def bufs():
        return dict(ws=torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda"), out=torch.empty(B * N, 64, device="cuda"),
                    lse=torch.empty(B * N, device="cuda"), dqkv=torch.empty(B * N, 192, device="cuda"))

    def run(b, st):
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(b["out"]), 64, _p(b["lse"]), 0.1, 7, None, _p(b["ws"]), st))
        _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(b["out"]), 64, _p(dO), 64, _p(b["lse"]), 0.1, 7, None, 0.125,
                                            _p(b["dqkv"]), 192, _p(b["ws"]), 1, st))

    ref = bufs()
    run(ref, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(S)]
    bs = [bufs() for _ in range(S)]
    for rep in range(5):
        for s, b in zip(streams, bs):
            with torch.cuda.stream(s):
                run(b, s.cuda_stream)
        torch.cuda.synchronize()
        for b in bs:
            for name in ("out", "lse", "dqkv"):
                assert torch.equal(b[name], ref[name]), (rep, name)


def test_training_kernels_are_not_disturbed_by_attention_on_another_stream():
    """The wider net around the packed-fp32 hazard (build rule): the kernels of a training episode's
    encoder pass and contrastive loss -- both kNN forms, EdgeConv training forward / backward, the point-wise and
    weight-gradient GEMMs, the BatchNorm reductions, the contrast kernels -- run on one stream while the attention kernels
    (bf16 x 3, then fp32) keep another stream busy: every feature and every gradient must equal the run alone, bit for
    bit.  (The batched trainer runs on ONE stream, so nothing executes beside the attention there; the multi-slot graph
    path of episode_graph.py does.)"""
    from types import SimpleNamespace
    from r3dfsseg_amd import _lib, contrast, synthetic as S, train_ops as T
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    from r3dfsseg_amd.ops import SegLayout, _p
    lib = _lib.load()
    cfg = S.workload_cfg("S")
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    m.load_state_dict(S.make_state_dict(cfg, 123))
    m.cuda().train()
    m.att_learner.dropout.p = 0.1
    data, _ = S.make_episode(cfg, seed=3, noise_ratio=0.2, train=True)
    ep = [t.cuda() for t in data]
    Sn, N = cfg["n_way"] * cfg["k_shot"], cfg["pc_npts"]
    x_all = torch.cat((ep[0].reshape(Sn, -1, N), ep[2]), 0).contiguous()
    B = x_all.shape[0]
    R = torch.randn(B * N, 192, device="cuda")
    qkv = torch.randn(B * N, 192, device="cuda")
    dO = torch.randn(B * N, 64, device="cuda")
    aws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
    out = torch.empty(B * N, 64, device="cuda")
    lse = torch.empty(B * N, device="cuda")
    dqkv = torch.empty(B * N, 192, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()

    def victim():
        with torch.cuda.stream(sa), torch.no_grad():
            c = SimpleNamespace(param_list=T.encoder_params(m), seg=SegLayout(1, Sn, B - Sn, N))
            feat = T.EncoderTrainFn.forward(c, x_all, m, 11)
            cc = SimpleNamespace()
            closs = contrast.ContrastFn.forward(cc, feat[:Sn * N], m.proj.weight, m.proj.bias, m, ep[1], ep[10])
            dc = contrast.ContrastFn.backward(cc, torch.ones((), device="cuda"))[:3]
            grads = T.EncoderTrainFn.backward(c, R)[3:]
            res = [feat.clone(), closs.clone()] + [g.clone() for g in dc] + [g.clone() for g in grads if g is not None]
        return res

    def attention(k):
        with torch.cuda.stream(sb):
            for _ in range(k):
                _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), 0.1, 7, None, _p(aws),
                                                       sb.cuda_stream))
                _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), 0.1, 7, None, 0.125,
                                                    _p(dqkv), 192, _p(aws), 1, sb.cuda_stream))

    T.update_running_stats = False
    before = lib.r3d_get_matrix_arith()
    try:
        for mode in (1, 0):
            _lib.check(lib.r3d_set_matrix_arith(mode))
            ref = victim()
            torch.cuda.synchronize()
            again = victim()
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(ref, again))  # (alone: reproducible)
            for rep in range(3):
                attention(60)
                got = victim()
                torch.cuda.synchronize()
                bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
                assert not bad, (mode, rep, bad)
    finally:
        T.update_running_stats = True
        _lib.check(lib.r3d_set_matrix_arith(before))
