"""Self-attention on the bf16 matrix core with fp32 operands cut into three bf16 pieces (csrc/common.h "bf16 x 3")
against the fp32 matrix core and a float64 reference of models/attention.py:43-46: both arithmetic modes must meet the
same 1e-4 bar, on ragged sizes too, and draw the same dropout mask."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ref(qkv, B, N, dO):
    q, k, v = (qkv[:, 64 * i:64 * (i + 1)].double().view(B, N, 64) for i in range(3))
    P = torch.softmax(q @ k.transpose(1, 2), -1)
    out = (P @ v).reshape(B * N, 64)
    g = dO.double().view(B, N, 64)
    dV = P.transpose(1, 2) @ g
    dP = g @ v.transpose(1, 2)
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    return out, torch.cat((dS @ k, dS.transpose(1, 2) @ q, dV), -1).reshape(B * N, 192)


def _run(lib, mode, qkv, B, N, dO, p_drop=0.0, seed=0):
    from r3dfsseg_amd import _lib
    from r3dfsseg_amd.ops import _p, _st
    before = lib.r3d_get_matrix_arith()
    _lib.check(lib.r3d_set_matrix_arith(mode))
    try:
        ws = torch.empty(lib.r3d_attention_ws_words(B, N), device="cuda")
        out = torch.empty(B * N, 64, device="cuda")
        lse = torch.empty(B * N, device="cuda")
        dqkv = torch.empty(B * N, 192, device="cuda")
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), p_drop, seed, None, _p(ws), _st()))
        _lib.check(lib.r3d_attention_bwd(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), p_drop, seed, None, 1.0,
                                         _p(dqkv), 192, _p(ws), _st()))
        # the backward on the forward's workspace (packed q | k | v reused): same result, bit for bit
        _lib.check(lib.r3d_attention_fwd_train(_p(qkv), 192, B, N, _p(out), 64, _p(lse), p_drop, seed, None, _p(ws), _st()))
        dqkv2 = torch.empty_like(dqkv)
        _lib.check(lib.r3d_attention_bwd_ws(_p(qkv), 192, B, N, _p(out), 64, _p(dO), 64, _p(lse), p_drop, seed, None, 1.0,
                                            _p(dqkv2), 192, _p(ws), 1, _st()))
        torch.cuda.synchronize()
        assert torch.equal(dqkv, dqkv2)
    finally:
        _lib.check(lib.r3d_set_matrix_arith(before))
    return out, dqkv


@pytest.mark.parametrize("B,N", [(12, 2048), (2, 2048), (3, 1000), (1, 77), (2, 4096)])
def test_attention_both_arithmetics_against_float64(B, N):
    from r3dfsseg_amd import _lib
    lib = _lib.load()
    torch.manual_seed(B * 10000 + N)
    qkv = torch.randn(B * N, 192, device="cuda")
    qkv[:, :64] *= 0.5
    dO = torch.randn(B * N, 64, device="cuda")
    want, want_d = _ref(qkv, B, N, dO)
    for mode in (0, 1):
        out, dqkv = _run(lib, mode, qkv, B, N, dO)
        err = ((out.double() - want).abs().max() / want.abs().max()).item()
        assert err <= TOL, (mode, err)
        for i in range(3):
            a, b = dqkv[:, 64 * i:64 * (i + 1)].double(), want_d[:, 64 * i:64 * (i + 1)]
            errd = ((a - b).abs().max() / b.abs().max()).item()
            assert errd <= TOL, (mode, i, errd)


def test_dropout_mask_is_the_same_in_both_arithmetics():
    from r3dfsseg_amd import _lib
    lib = _lib.load()
    B, N = 3, 1000
    torch.manual_seed(5)
    qkv = torch.randn(B * N, 192, device="cuda")
    dO = torch.randn(B * N, 64, device="cuda")
    o0, d0 = _run(lib, 0, qkv, B, N, dO, p_drop=0.1, seed=1234)
    o1, d1 = _run(lib, 1, qkv, B, N, dO, p_drop=0.1, seed=1234)
    # a different mask would move an output by ~ p / sqrt(N) of its size, three orders above this bound
    assert ((o0 - o1).abs().max() / o0.abs().max()).item() <= 1e-5
    assert ((d0 - d1).abs().max() / d0.abs().max()).item() <= 1e-4
    o2, _ = _run(lib, 1, qkv, B, N, dO, p_drop=0.1, seed=1235)
    assert ((o2 - o1).abs().max() / o0.abs().max()).item() > 1e-3


def test_arith_mode_is_validated():
    from r3dfsseg_amd import _lib
    lib = _lib.load()
    import os
    before = lib.r3d_get_matrix_arith()
    assert before == (0 if os.environ.get("R3D_MATRIX_ARITH") == "fp32" else 1)  # bf16 x 3 unless the environment says fp32
    assert lib.r3d_set_matrix_arith(7) != 0
    assert "r3d_set_matrix_arith" in lib.r3d_last_error_string().decode()
    assert lib.r3d_get_matrix_arith() == before
