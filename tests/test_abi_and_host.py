"""CPU-side checks: the C-ABI library builds, loads and exports every symbol declared in
include/r3d.h (no compute without a GPU); host-side logic of the interface mirror."""
import ctypes
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from r3dfsseg_amd import _lib, synthetic as S


def test_library_exports_every_declared_symbol():
    from r3dfsseg_amd import build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _lib.header_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib._SIGS) == set(declared), set(_lib._SIGS) ^ set(declared)
    lib.r3d_abi_version.restype = ctypes.c_int
    assert lib.r3d_abi_version() == 4
    lib.r3d_head_desc_words.restype = ctypes.c_int
    assert lib.r3d_head_desc_words() == 32
    lib.r3d_lp_ws_words.restype = ctypes.c_long
    assert lib.r3d_lp_ws_words(4396, 201) > 0


def test_loading_a_library_of_another_abi_version_is_refused(monkeypatch):
    """A stale libr3d_hip.so would be called with this package's argument lists (round 4 added arguments to two entry
    points): load() compares r3d_abi_version() with the version the bindings were written for and says how to rebuild."""
    from r3dfsseg_amd import build
    build.build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="C ABI version 4.*rebuild"):
        _lib.load()
    monkeypatch.setattr(_lib, "ABI_VERSION", 4)
    assert _lib.load().r3d_abi_version() == 4


def test_binding_signatures_match_the_header():
    """Every prototype of include/r3d.h against the ctypes signature the binding declares for it: argument count, and per
    argument the class of its type (pointer / int / long / float / double / unsigned).  A binding that has drifted from the
    header passes a long where the library reads an int -- silently, on this ABI."""
    import re
    txt = open(_lib.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    protos = re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(r3d_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
    assert len(protos) >= 80, len(protos)

    def cls(decl):
        d = " ".join(decl.split())
        if "*" in d:
            return "ptr"
        base = d.rsplit(" ", 1)[0] if " " in d else d  # drop the parameter name
        for key, name in (("unsigned", "uint"), ("double", "double"), ("float", "float"), ("long", "long"), ("int", "int")):
            if key in base:
                return name
        raise AssertionError("unparsed parameter %r" % decl)

    ctype_cls = {_lib.c_f: "ptr", _lib.c_i: "int", _lib.c_l: "long", _lib.c_fl: "float", _lib.c_d: "double", _lib.c_u: "uint",
                 ctypes.c_char_p: "ptr"}
    seen = set()
    for ret, name, args in protos:
        seen.add(name)
        res, argt = _lib._SIGS[name]
        params = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        assert len(params) == len(argt), (name, len(params), len(argt))
        for i, (p_, t_) in enumerate(zip(params, argt)):
            want = "ptr" if hasattr(t_, "contents") else ctype_cls[t_]  # (ctypes.POINTER(...) types: host pointers)
            assert cls(p_) == want, (name, i, p_, t_)
        assert cls(ret + " x") == ctype_cls[res], (name, ret, res)
    assert seen == set(_lib._SIGS), seen ^ set(_lib._SIGS)


def test_abi_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (error convention of include/r3d.h)."""
    lib = _lib.load()
    rc = lib.r3d_knn_topk(None, 9, None, 1, 64, 9, 20, 0, None, None, None, None, None, None, None)
    assert rc != 0 and b"null" in lib.r3d_last_error_string()
    rc = lib.r3d_edgeconv_fwd(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8),
                              ctypes.c_void_p(8), ctypes.c_void_p(8), 64, 1, 102, 20, None, None)
    assert rc != 0 and b"multiple" in lib.r3d_last_error_string()
    rc = lib.r3d_graph_set_lp_budget(None, None, 16, None)
    assert rc != 0 and b"r3d_graph_set_lp_budget" in lib.r3d_last_error_string()


def test_state_dict_names_match_reference_contract():
    """Key names and shapes of SURVEY.md 8b (verified there against the reference DGCNN)."""
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    cfg = S.make_cfg()
    m = MPTI_SelfAtten(SimpleNamespace(**cfg))
    sd = m.state_dict()
    want = S.make_state_dict(cfg)
    assert list(sd.keys()) == list(want.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(want[k].shape), k
    assert sd["encoder.edge_convs.0.layer.0.weight"].shape == (64, 18, 1, 1)
    assert sd["encoder.conv.layer.3.weight"].shape == (256, 512, 1)
    assert sd["att_learner.q_map.weight"].shape == (64, 256, 1)
    assert sd["proj.weight"].shape == (128, 192)
    n_train = sum(p.numel() for p in m.parameters())
    assert n_train == 376896  # 261504 + 41536 + 49152 + 24704
    m.load_state_dict(want)


def test_unsupported_configs_raise():
    from r3dfsseg_amd.mpti import MPTI_SelfAtten
    with pytest.raises(NotImplementedError):
        MPTI_SelfAtten(SimpleNamespace(**S.make_cfg(edgeconv_widths=[[64, 32]] * 3)))
    with pytest.raises(NotImplementedError):
        MPTI_SelfAtten(SimpleNamespace(**S.make_cfg(n_way=8)))  # (up to 7 ways: two planes of four label columns)


def test_product_never_imports_oracle():
    import r3dfsseg_amd
    root = os.path.dirname(r3dfsseg_amd.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libr3d_oracle" not in txt, f


def test_synthetic_episode_contract():
    cfg = S.workload_cfg("S")
    data, classes = S.make_episode(cfg, 1, noise_ratio=0.4, train=True)
    assert len(data) == 11
    sx, sy, qx, qy = data[:4]
    assert sx.shape == (2, 5, 9, 2048) and sy.shape == (2, 5, 2048) and sy.dtype == torch.int32
    assert qx.shape == (2, 9, 2048) and qy.dtype == torch.int64 and qy.max() <= 2
    gsy, flag = data[6], data[10]
    assert (gsy[:, 3:].sum() == 0) and (gsy[:, :3] == sy[:, :3]).all()  # 2 of 5 shots are noise
    assert flag.shape == (2, 5) and (flag[:, 3:] == 99).all()
    d2, _ = S.make_episode(cfg, 1, noise_ratio=0.4, train=True)
    assert all(torch.equal(a, b) for a, b in zip(data, d2))  # same seed, same bytes


def test_no_packed_fp32_arithmetic_in_the_device_code(tmp_path):
    """The library is built without v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (r3dfsseg_amd/build.py says why: wrong
    results in a wave running them beside bf16-MFMA-dense waves, measured on MI355X).  Disassemble every gfx950 code
    object of the built library and look."""
    import glob
    import shutil
    import subprocess
    from r3dfsseg_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is not here")
    so = shutil.copy(_lib.LIB_PATH, str(tmp_path / "lib.so"))
    subprocess.run([objdump, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert len(objs) >= 8, objs
    n_mfma = 0
    for o in objs:
        dis = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        n_mfma += dis.count("v_mfma_")
        for op in ("v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32"):
            assert op not in dis, (op, os.path.basename(o))
    assert n_mfma > 100  # it really was the device code


def test_lane_writes_keep_their_distance_from_the_mask_they_read(tmp_path):
    """The kNN filter pass collects ballot masks with blocks of 16 `v_writelane_b32 v, sN, lane` from inline assembly,
    which the compiler's hazard recogniser does not see; on gfx950 such a write reads a STALE sN when it follows the
    vector instruction that wrote sN directly (csrc/knn.hip, profiles/r04_experiments.md section 6).  Look at the built
    code: every such block (8 or more lane writes in a row -- the compiler's own SGPR spills come in ones and twos and
    carry its own wait states) starts behind `s_nop 4`, the distance that was measured to be enough."""
    import glob
    import re
    import shutil
    import subprocess
    from r3dfsseg_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or the built library is not here")
    so = shutil.copy(_lib.LIB_PATH, str(tmp_path / "lib.so"))
    subprocess.run([objdump, "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blocks = 0
    for o in glob.glob(str(tmp_path / "lib.so.*gfx950")):
        dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
        ins = [l.split("//")[0].strip() for l in dis.splitlines() if l.startswith("\t")]
        i = 0
        while i < len(ins):
            j = i
            while j < len(ins) and re.match(r"v_writelane_b32 v\d+, (?:s\d+|vcc_lo|vcc_hi), \d+$", ins[j]):
                j += 1
            if j - i >= 8:
                blocks += 1
                assert i > 0 and ins[i - 1] == "s_nop 4", (os.path.basename(o), ins[max(0, i - 3):i + 1])
            i = max(j, i + 1)
    assert blocks >= 4  # two blocks per sub-tile in every instantiation of the filter pass
