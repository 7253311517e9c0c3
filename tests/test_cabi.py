"""The C-ABI library loads and exports every symbol include/cxrk.h declares, with matching argument counts
(no compute calls: this runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "cxrk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(cxrk_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_header_declares_the_hot_path_ops():
    fns = _header_functions()
    for op in ("gemm_f32", "conv_bn_act_fwd", "conv_bn_act_bwd_data", "conv_bn_act_bwd_params", "maxpool_fwd", "spatial_mean_fwd",
               "embed_ln_fwd", "attn_fwd", "attn_bwd", "residual_ln_fwd", "residual_ln_bwd", "l2norm_fwd", "infonce_row_lse",
               "pairwise_cosine_fwd", "bce_posneg_fwd_bwd", "adam_fused", "weight_reset"):
        assert "cxrk_" + op in fns, op


def test_library_exports_every_declared_symbol():
    from incremental_multimodal_medical_learning_ii_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    fns = _header_functions()
    assert len(fns) >= 40
    for name, nargs in fns.items():
        assert hasattr(lib, name), f"{name} declared in include/cxrk.h but not exported by libcxrk.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, len(_lib.SIGNATURES[name][1]), nargs)
    assert set(_lib.SIGNATURES) == set(fns), set(_lib.SIGNATURES) ^ set(fns)
    assert lib.cxrk_version().decode().endswith("gfx950")


def test_workspace_size_queries_are_pure():
    from incremental_multimodal_medical_learning_ii_amd import _lib
    lib = _lib.load()
    assert lib.cxrk_gemm_splitk_ws_bytes(128, 256, 4) == 4 * 128 * 256 * 4
    assert lib.cxrk_gemm_splitk_ws_bytes(128, 256, 1) == 0
    assert lib.cxrk_colsum_ws_bytes(10_000, 768) > 0
    assert lib.cxrk_conv_wgrad_ws_bytes(2, 56, 56, 64, 64, 3, 3, 1, 1) >= 64 * 9 * 64 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from incremental_multimodal_medical_learning_ii_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CxrkError):
        _lib.load()
