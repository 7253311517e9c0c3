"""ctypes binding of libcxrk.so (the C-ABI declared in include/cxrk.h).

The library is the product: there is NO fallback.  If it is missing, or a call returns an error code, the caller
gets an exception (`CxrkError`), never a silently different code path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_void_p
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcxrk.so")

P, I, L, F, Z = c_void_p, c_int, c_long, c_float, c_size_t

# name -> (restype, argtypes).  Must list every symbol of include/cxrk.h (tests/test_cabi.py checks it).
SIGNATURES: Dict[str, Tuple[object, List[object]]] = {
    "cxrk_gemm_splitk_ws_bytes": (Z, [I, I, I]),
    "cxrk_gemm_f32": (I, [I, I, I, I, I, P, L, P, L, P, L, P, P, L, P, L, I, P, L, I, F, I, I, P, Z, P]),
    "cxrk_gemm_pl": (I, [I, I, I, I, I, P, L, L, P, L, L, P, P, L, L, P, P, P, L, L, P, L, I, P, L, P, L, P, L, I, F, I, I, P, I, P, Z, P]),
    "cxrk_gemm_pl_colsum_ws_bytes": (Z, [I, I]),
    "cxrk_split_planes": (I, [P, L, P, L, P]),
    "cxrk_merge_planes": (I, [P, L, L, P, P]),
    "cxrk_colsum_pl": (I, [P, L, L, L, I, P, F, I, P, Z, P]),
    "cxrk_colsum_ws_bytes": (Z, [L, I]),
    "cxrk_colsum": (I, [P, L, L, I, P, F, I, P, Z, P]),
    "cxrk_colvar": (I, [P, L, L, L, I, P, P, F, P, Z, P]),
    "cxrk_bn_fold": (I, [P, P, P, P, P, F, I, I, I, I, P, P, P, P, P]),
    "cxrk_conv_bn_act_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "cxrk_bn_fold_pl": (I, [P, P, P, P, P, F, I, I, I, I, P, L, P, P, P, P]),
    "cxrk_conv_bn_act_fwd_pl": (I, [P, L, P, L, I, P, P, L, P, L, P, I, I, I, I, I, I, I, I, I, I, P]),
    "cxrk_conv_bwd_data_colsum_ws_bytes": (Z, [I, I, I, I, I]),
    "cxrk_conv_bn_act_bwd_data": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, Z, P]),
    "cxrk_conv_bn_act_bwd_data_pl": (I, [P, L, P, L, P, L, P, P, L, I, I, I, I, I, I, I, I, I, P, P, Z, P]),
    "cxrk_conv_bn_act_bwd_data_pl_s2res": (I, [P, L, P, L, P, L, P, P, L, I, I, I, I, I, I, I, I, I, P, P, Z, P]),
    "cxrk_conv_wgrad_ws_bytes": (Z, [I, I, I, I, I, I, I, I, I]),
    "cxrk_conv_bn_act_bwd_params": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, Z, P]),
    "cxrk_conv_bn_act_bwd_params_pl": (I, [P, L, P, L, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, Z, P]),
    "cxrk_colstats": (I, [P, L, L, I, P, P, F, P, Z, P]),
    "cxrk_bn_train_fwd_coeffs": (I, [P, P, P, P, F, L, F, P, P, P, P, P, I, P]),
    "cxrk_bn_apply": (I, [P, L, P, P, P, L, P, L, P, L, I, I, P]),
    "cxrk_coldot_ws_bytes": (Z, [L, I]),
    "cxrk_coldot": (I, [P, L, P, L, P, L, I, P, P, Z, P]),
    "cxrk_bn_train_bwd_coeffs": (I, [P, P, P, P, P, L, P, P, P, P, P, I, I, P]),
    "cxrk_bn_train_dz": (I, [P, L, P, L, P, P, P, P, L, L, I, P]),
    "cxrk_nchw_to_nhwc": (I, [P, P, I, I, I, I, I, P]),
    "cxrk_nhwc_to_nchw": (I, [P, P, I, I, I, I, P]),
    "cxrk_maxpool_fwd": (I, [P, P, P, I, I, I, I, P]),
    "cxrk_maxpool_bwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "cxrk_maxpool_fwd_pl": (I, [P, L, P, L, P, I, I, I, I, P]),
    "cxrk_maxpool_bwd_pl": (I, [P, L, P, P, P, I, I, I, I, P]),
    "cxrk_spatial_mean_fwd": (I, [P, P, I, I, I, P]),
    "cxrk_spatial_mean_bwd": (I, [P, P, I, I, I, P]),
    "cxrk_spatial_mean_bwd_pl": (I, [P, P, P, L, I, I, I, P]),
    "cxrk_embed_ln_fwd": (I, [P, P, P, P, P, P, F, L, I, I, P, L, P, P, P]),
    "cxrk_residual_ln_fwd": (I, [P, P, P, P, F, L, I, P, L, P, P, P]),
    "cxrk_residual_ln_bwd_ws_bytes": (Z, [L, I]),
    "cxrk_residual_ln_bwd": (I, [P, P, P, P, L, I, P, P, L, P, P, I, P, I, P, Z, P]),
    "cxrk_planes_add_rows": (I, [P, L, L, I, P, L, P]),
    "cxrk_attn_fwd": (I, [P, P, I, I, I, I, P, L, P, P]),
    "cxrk_attn_bwd_ws_bytes": (Z, [I, I, I, I]),
    "cxrk_attn_bwd": (I, [P, P, P, I, I, I, I, P, L, P, Z, P]),
    "cxrk_embed_bwd_ws_bytes": (Z, [L, I]),
    "cxrk_embed_bwd": (I, [P, P, L, I, P, P, Z, P]),
    "cxrk_gelu_bwd": (I, [P, P, L, P, P]),
    "cxrk_l2norm_fwd": (I, [P, L, I, F, P, L, P, P]),
    "cxrk_l2norm_bwd": (I, [P, P, L, P, L, I, P, P]),
    "cxrk_infonce_row_lse": (I, [P, L, I, I, I, P, P, P, F, I, P]),
    "cxrk_infonce_grad_inplace": (I, [P, L, I, I, I, P, P, P]),
    "cxrk_pairwise_cosine_fwd": (I, [P, P, L, I, I, P, P, P, P]),
    "cxrk_pairwise_cosine_bwd_ws_bytes": (Z, [L, I, I]),
    "cxrk_pairwise_cosine_bwd": (I, [P, P, P, P, P, P, L, I, I, P, P, I, P, Z, P]),
    "cxrk_pairwise_cosine_max_fwd": (I, [P, P, L, I, I, I, P, P, P, P, P, P, P]),
    "cxrk_pairwise_cosine_max_bwd": (I, [P, P, P, P, P, P, P, L, I, I, I, P, P, I, P, Z, P]),
    "cxrk_patch_similarity": (I, [P, P, L, I, P, P]),
    "cxrk_bce_posneg_ws_bytes": (Z, []),
    "cxrk_bce_posneg_fwd_bwd": (I, [P, P, L, I, I, I, P, P, P, P, Z, P]),
    "cxrk_eval_score": (I, [P, L, I, I, P, P, P]),
    "cxrk_group_mean_fwd": (I, [P, I, I, I, P, P]),
    "cxrk_group_mean_bwd": (I, [P, I, I, I, P, P]),
    "cxrk_scale_mask": (I, [P, P, P, F, L, P, P]),
    "cxrk_adam_fused": (I, [P, P, P, P, L, F, F, F, F, F, I, F, P]),
    "cxrk_sgd": (I, [P, P, L, F, F, F, P]),
    "cxrk_weight_reset_ws_bytes": (Z, []),
    "cxrk_weight_reset": (I, [P, P, L, F, P, P, Z, P]),
    "cxrk_gemm_wgrad_splitk": (I, [I, I, I, I]),
    "cxrk_gemm_wide_tile": (I, [I, I, L, I, I]),
    "cxrk_set_precision": (I, [I]),
    "cxrk_get_precision": (I, []),
    "cxrk_set_wide_mode": (I, [I]),
    "cxrk_version": (c_char_p, []),
}

_ERRORS = {-1: "bad argument / shape / alignment", -2: "workspace too small", -3: "kernel launch failed",
           -4: "unsupported shape"}


class CxrkError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load libcxrk.so (built by `__graft_entry__.build()` / `make -C csrc`).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CxrkError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    mode = os.environ.get("CXRK_PRECISION", "fp32")
    if mode not in PRECISIONS:
        raise CxrkError(f"CXRK_PRECISION={mode!r}: expected one of {sorted(PRECISIONS)}")
    lib.cxrk_set_precision(PRECISIONS[mode])
    return lib


PRECISIONS = {"fp32": 0, "split_bf16": 1}


def set_precision(mode: str) -> None:
    """Contraction precision of every GEMM / convolution: "fp32" (exact fp32 MFMA, default) or "split_bf16"
    (3x bf16 MFMA per product, fp32 accumulate, ~2^-16 relative per product)."""
    if mode not in PRECISIONS:
        raise ValueError(f"precision {mode!r}: expected one of {sorted(PRECISIONS)}")
    check(load().cxrk_set_precision(PRECISIONS[mode]), "cxrk_set_precision")


def get_precision() -> str:
    v = load().cxrk_get_precision()
    return {b: a for a, b in PRECISIONS.items()}[v]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _ERRORS.get(rc, "unknown error")
        if rc in (-1, -4):
            raise ValueError(f"{what}: {msg} (cxrk code {rc})")
        raise CxrkError(f"{what}: {msg} (cxrk code {rc})")
