"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every entry
point include/vitssl_hip.h declares (no compute calls here)."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    import vitssl_hip
    return vitssl_hip


def test_exports_every_header_symbol(built):
    syms = built.header_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(built.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/vitssl_hip.h but not exported"
    from vitssl_hip import _lib
    declared = set(syms) - {"vitssl_last_error", "vitssl_version", "vitssl_gemm_tn_workspace_floats",
                            "vitssl_gemm_fp8_tn_workspace_floats", "vitssl_embed_bwd_workspace_floats",
                            "vitssl_gemm_tn_batch_workspace_floats", "vitssl_gemm_fp8_tn_batch_workspace_floats", "vitssl_dino_loss_workspace_floats",
                            "vitssl_get_reserved_cus", "vitssl_debug_last_nt_grid", "vitssl_debug_last_attn_fwd_grid"}
    assert declared == set(_lib.PROTOTYPES), "Python prototypes out of sync with the header"
    assert built.lib().vitssl_version() >= 2


def test_no_cpu_fallback(built):
    import torch
    from vitssl_hip import ops, _lib
    x = torch.zeros(4, 64)
    with pytest.raises(_lib.VitsslError):
        ops.cast_bf16(x, torch.zeros(4, 64, dtype=torch.bfloat16))


def test_argument_errors_are_reported(built):
    from vitssl_hip import _lib
    g = _lib.Gemm()
    rc = built.lib().vitssl_gemm_bf16_nt(ctypes.byref(g), None)
    assert rc == -1 and b"null operand" in built.lib().vitssl_last_error()
