"""ctypes binding of libr3d_hip.so (the C ABI declared in include/r3d.h).

This is the stub a maintainer of the reference would add next to models/ to call the
MI355X kernels.  There is NO fallback: if the shared library is missing or a symbol is
absent, importing an op raises -- the product path never computes on the CPU.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R3D_LIB") or os.path.join(_HERE, "libr3d_hip.so")  # R3D_LIB: probe builds (tools/probe)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "r3d.h")

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i = ctypes.c_int
c_l = ctypes.c_long
c_fl = ctypes.c_float
c_d = ctypes.c_double
c_u = ctypes.c_uint

_SIGS = {
    "r3d_last_error_string": (ctypes.c_char_p, []),
    "r3d_abi_version": (c_i, []),
    "r3d_cm_to_pm": (c_i, [c_f, c_i, c_i, c_i, c_f, c_l, c_f]),
    "r3d_pm_to_cm": (c_i, [c_f, c_l, c_i, c_i, c_i, c_f, c_f]),
    "r3d_pm_to_cm_pitched": (c_i, [c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f]),
    "r3d_cm_pitch": (c_l, [c_i]),
    "r3d_copy_cols": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_f]),
    "r3d_sqnorm": (c_i, [c_f, c_l, c_l, c_i, c_f, c_f]),
    "r3d_knn_topk": (c_i, [c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "r3d_knn_split_ws_words": (c_l, [c_i, c_i, c_i]),
    "r3d_knn_topk_split": (c_i, [c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_f]),
    "r3d_pointwise_conv": (c_i, [c_f, c_l, c_f, c_l, c_i, c_i, c_f, c_f, c_i, c_f, c_l, c_f]),
    "r3d_edgeconv_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_f]),
    "r3d_attention_ws_words": (c_l, [c_i, c_i]),
    "r3d_knn_norm_ws_words": (c_l, [c_i, c_i]),
    "r3d_attention_fwd": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_f, c_f]),
    "r3d_head_desc_words": (c_i, []),
    "r3d_debug_poison_lds": (c_i, [c_u, c_f, c_f]),
    "r3d_debug_set_cg_spmv_lds_min_blocks": (c_i, [c_i]),
    "r3d_debug_set_gemm_bx3": (c_i, [c_i]),
    "r3d_set_matrix_arith": (c_i, [c_i]),
    "r3d_get_matrix_arith": (c_i, []),
    "r3d_head_max_k": (c_i, []),
    "r3d_fps_sample_count_table": (c_i, [c_i, c_i, c_f, c_f]),
    "r3d_head_proto_ws_words": (c_l, [c_i, c_i, c_i]),
    "r3d_head_proto_ws_offsets": (c_i, [c_i, c_i, c_i, ctypes.POINTER(c_l)]),
    "r3d_head_prototypes": (c_i, [c_f, c_f, c_f, c_l, c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_l,
                                  c_f, c_f, c_f, c_f, c_f, c_l, c_i, c_f]),
    "r3d_lp_ws_words": (c_l, [c_i, c_i]),
    "r3d_lp_ws_offsets": (c_i, [c_i, c_i, ctypes.POINTER(c_l)]),
    "r3d_label_propagate": (c_i, [c_f, c_l, c_i, c_f, c_i, c_f, c_f, c_f, c_i, c_fl, c_fl, c_i, c_fl, c_f, c_f,
                                  c_l, c_f, c_f]),
    "r3d_graph_set_lp_budget": (c_i, [c_f, c_f, c_i, ctypes.POINTER(c_i)]),
    "r3d_pointwise_conv_acc": (c_i, [c_f, c_l, c_f, c_l, c_i, c_i, c_f, c_f, c_i, c_f, c_l, c_f]),
    "r3d_edgeconv_train_fwd_minmax": (c_i, [c_f, c_f, c_f, c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "r3d_edge_select": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f]),
    "r3d_pointwise_conv_stats_ws_words": (c_l, [c_l, c_i]),
    "r3d_pointwise_conv_stats": (c_i, [c_f, c_l, c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_f, c_f]),
    "r3d_pointwise_conv_stats2": (c_i, [c_f, c_l, c_f, c_l, c_i, c_i, c_f, c_l, c_l, c_f, c_f, c_f, c_f]),
    "r3d_colreduce": (c_i, [c_f, c_i, c_i, c_f, c_f]),
    "r3d_colstats_ws_words": (c_l, [c_l, c_i]),
    "r3d_colstats": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_f]),
    "r3d_bn_fold": (c_i, [c_f, c_d, c_i, c_f, c_f, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_f]),
    "r3d_bn_running_update": (c_i, [c_f, c_i, c_l, c_i, c_fl, c_f, c_f, c_f, c_f]),
    "r3d_affine_act": (c_i, [c_f, c_l, c_l, c_i, c_f, c_f, c_i, c_f, c_l, c_f]),
    "r3d_bn_bwd_apply": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_d, c_f, c_l, c_f]),
    "r3d_gemm_tn_ws_words": (c_l, [c_l, c_i, c_i]),
    "r3d_gemm_tn": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_i, c_fl, c_f, c_i, c_f, c_f]),
    "r3d_add_cols": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_f]),
    "r3d_edgeconv_train_ws_words": (c_l, [c_i, c_i]),
    "r3d_edge_stats1": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "r3d_edge_reverse_ws_words": (c_l, [c_i, c_i, c_i]),
    "r3d_edge_reverse": (c_i, [c_f, c_i, c_i, c_i, c_f, c_l, c_f]),
    "r3d_edgeconv_bwd": (c_i, [c_f] * 11 + [c_l, c_f, c_f, c_l, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "r3d_attention_fwd_train": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_fl, c_u, c_f, c_f, c_f]),
    "r3d_attention_bwd": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_fl, c_u, c_f, c_fl, c_f, c_l, c_f, c_f]),
    "r3d_attention_bwd_ws": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_fl, c_u, c_f, c_fl, c_f, c_l, c_f, c_i, c_f]),
    "r3d_ce_grad": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "r3d_label_propagate_bwd": (c_i, [c_f, c_l, c_i, c_i, c_f, c_f, c_f, c_i, c_fl, c_fl, c_i, c_fl, c_f, c_f, c_l, c_f, c_l, c_f, c_f]),
    "r3d_head_prototypes_bwd": (c_i, [c_f, c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_l, c_f, c_l, c_f]),
    "r3d_contrast_ws_words": (c_l, [c_i, c_i, c_i]),
    "r3d_contrast_fwd": (c_i, [c_f, c_l, c_i, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_fl, c_f, c_f, c_l, c_f]),
    "r3d_contrast_bwd": (c_i, [c_i, c_i, c_i, c_i, c_f, c_f, c_l, c_f, c_f, c_f, c_f]),
    "r3d_train_metrics": (c_i, [c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "r3d_clean_ws_words": (c_l, [c_i, c_i]),
    "r3d_clean_shot_detect": (c_i, [c_f, c_l, c_i, c_f, c_i, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "r3d_protonet_head": (c_i, [c_f, c_l, c_f, c_l, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, c_f]),
    "r3d_miou_accumulate": (c_i, [c_f, c_f, c_l, c_f, c_i, c_i, c_f, c_f]),
    "r3d_query_logits_ce": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f]),
    # ---- ABI version 3: segments (training encoder) and batches of episodes (head)
    "r3d_knn_topk_batched": (c_i, [c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_f, c_l,
                                   c_f]),
    "r3d_knn_bf_ws_words": (c_l, [c_i, c_i, c_i]),
    "r3d_debug_set_knn_bf16_threshold": (c_i, [c_i]),
    "r3d_debug_set_knn_bf16_filter": (c_i, [c_i]),
    "r3d_set_wpack_in_capture": (c_i, [c_i]),
    "r3d_pointwise_conv_stats_seg": (c_i, [c_f, c_l, c_f, c_l, c_i, c_i, c_f, c_l, c_l, c_l, c_f, c_f, c_f]),
    "r3d_colreduce_seg": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "r3d_colstats_seg_ws_words": (c_l, [c_l, c_i, c_l, c_l]),
    "r3d_colstats_seg": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_l, c_l, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_f, c_f]),
    "r3d_bn_fold_seg": (c_i, [c_f, c_i, c_d, c_d, c_i, c_f, c_f, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_f, c_f, c_l, c_f]),
    "r3d_affine_act_seg": (c_i, [c_f, c_l, c_l, c_i, c_l, c_l, c_f, c_f, c_l, c_i, c_f, c_l, c_f]),
    "r3d_bn_bwd_apply_seg": (c_i, [c_f, c_l, c_f, c_l, c_l, c_i, c_l, c_l, c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_d, c_d, c_f, c_l, c_f]),
    "r3d_attention_ws_words_ep": (c_l, [c_i, c_i, c_i]),
    "r3d_attention_fwd_train_ep": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_fl, c_u, c_f, c_i, c_f, c_f]),
    "r3d_attention_bwd_ep": (c_i, [c_f, c_l, c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_fl, c_u, c_f, c_i, c_fl, c_f, c_l, c_f, c_i, c_f]),
    "r3d_head_prototypes_batched": (c_i, [c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_l, c_l, c_f, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_i,
                                          c_f, c_l, c_l, c_f, c_f, c_l, c_f, c_l, c_f, c_l, c_f, c_l, c_l, c_i, c_f]),
    "r3d_head_prototypes_bwd_batched": (c_i, [c_i, c_f, c_l, c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_l, c_f, c_l, c_f, c_l, c_f, c_l,
                                              c_f, c_l, c_l, c_f, c_l, c_l, c_f]),
    "r3d_graph_weights_verify_words": (c_l, [c_i, c_i]),
    "r3d_graph_weights_verify": (c_i, [c_i, c_f, c_l, c_i, c_f, c_l, c_i, c_i, c_fl, c_f, c_l, c_l, c_f, c_f, c_f]),
    "r3d_label_propagate_solve_batched": (c_i, [c_i, c_f, c_f, c_l, c_i, c_i, c_fl, c_i, c_fl, c_f, c_f, c_l, c_l, c_f, c_l, c_f]),
    "r3d_label_propagate_batched": (c_i, [c_i, c_f, c_l, c_i, c_f, c_i, c_f, c_f, c_f, c_l, c_i, c_fl, c_fl, c_i, c_fl, c_f, c_f,
                                          c_l, c_l, c_f, c_l, c_f]),
    "r3d_label_propagate_bwd_batched": (c_i, [c_i, c_f, c_l, c_i, c_i, c_f, c_f, c_f, c_l, c_i, c_fl, c_fl, c_i, c_fl, c_f, c_f,
                                              c_l, c_f, c_l, c_l, c_f, c_l, c_f]),
    "r3d_ce_grad_batched": (c_i, [c_i, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "r3d_query_logits_ce_batched": (c_i, [c_i, c_f, c_l, c_f, c_l, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f]),
    "r3d_contrast_fwd_batched": (c_i, [c_i, c_f, c_l, c_l, c_i, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_fl, c_f, c_f, c_l, c_l, c_f]),
    "r3d_contrast_bwd_batched": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_l, c_l, c_f, c_f, c_f, c_l, c_f]),
    "r3d_train_metrics_batched": (c_i, [c_i, c_f, c_f, c_f, c_i, c_f, c_l, c_f, c_l, c_f, c_l, c_f, c_l, c_f, c_i, c_i, c_i, c_f, c_f]),
    "r3d_clean_shot_detect_batched": (c_i, [c_i, c_f, c_l, c_l, c_i, c_f, c_i, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_l, c_f]),
}

_lib = None


def header_symbols():
    """Function names declared in include/r3d.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(r3d_[a-z0-9_]+)\s*\(", txt)))


ABI_VERSION = 4  # include/r3d.h; 4 (round 4): r3d_edge_stats1(+esum), r3d_edgeconv_bwd(+zwin, esum)


def load():
    """Load the shared library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one already mapped when our library's
    # libamdhip64 dependency is resolved (two runtimes in one process see no device)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "r3dfsseg_amd: %s not found -- build it with `python -m r3dfsseg_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    cdll = ctypes.CDLL(LIB_PATH)
    lib = _Bound()
    lib._cdll = cdll
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError:
            raise RuntimeError("r3dfsseg_amd: symbol %s missing from %s" % (name, LIB_PATH))
        fn.restype = res
        fn.argtypes = args
        setattr(lib, name, fn)
    if lib.r3d_abi_version() != ABI_VERSION:  # a stale build would be CALLED with this round's argument lists
        raise RuntimeError("r3dfsseg_amd: %s has C ABI version %d, this package binds version %d -- rebuild it "
                           "(`python -m r3dfsseg_amd.build --force`)" % (LIB_PATH, lib.r3d_abi_version(), ABI_VERSION))
    _lib = lib
    mode = os.environ.get("R3D_MATRIX_ARITH")  # "fp32" | "bf16x3": see r3d_set_matrix_arith in include/r3d.h
    if mode is not None:
        if mode not in ("fp32", "bf16x3"):
            raise RuntimeError("R3D_MATRIX_ARITH=%r: expected fp32 or bf16x3" % mode)
        check(lib.r3d_set_matrix_arith(1 if mode == "bf16x3" else 0))
    mask = os.environ.get("R3D_GEMM_BX3")  # tuning knob: r3d_debug_set_gemm_bx3 in include/r3d.h
    if mask is not None:
        check(lib.r3d_debug_set_gemm_bx3(int(mask)))
    return lib


class _Bound:
    """The bound entry points as plain attributes (one per symbol of include/r3d.h)."""


_call_log = None  # list collecting (fn, args) of the library calls of one timed region (bench.py's roofline leg)


def record_calls(on):
    """Route every entry point through a recorder (on=True) or back to the raw ctypes functions (on=False).
    While on, calls made with `_call_log` set to a list are appended to it, so that a timed region can be
    launched again back to back (ops.KernelTimer with repeat > 0)."""
    lib = load()
    for name in _SIGS:
        raw = getattr(lib._cdll, name)
        if not on:
            setattr(lib, name, raw)
            continue

        def wrapper(*args, _raw=raw):
            rc = _raw(*args)
            if _call_log is not None:
                _call_log.append((_raw, args))
            return rc
        setattr(lib, name, wrapper)


def check(rc):
    if rc != 0:
        raise RuntimeError("r3d: " + load().r3d_last_error_string().decode())
