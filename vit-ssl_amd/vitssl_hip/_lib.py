"""ctypes binding of libvitssl_hip.so (C ABI declared in include/vitssl_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises."""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITSSL_LIB") or os.path.join(_HERE, "libvitssl_hip.so")   # VITSSL_LIB: developer override (kernel A/B builds)
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "vitssl_hip.h"))


class VitsslError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_uint64)]


class Embed(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("mask_token", C.c_void_p), ("pos", C.c_void_p),
                ("tokens", C.c_int), ("out_tokens", C.c_int), ("tok_offset", C.c_int)]


class TnJob(C.Structure):
    """vitssl_tn_job_t"""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("N1", C.c_int), ("N2", C.c_int)]


class Fp8TnJob(C.Structure):
    """vitssl_fp8_tn_job_t"""
    _fields_ = [("A8", C.c_void_p), ("B8", C.c_void_p), ("C", C.c_void_p), ("N1", C.c_int), ("N2", C.c_int),
                ("alpha", C.c_void_p), ("alpha2", C.c_void_p)]


class Gemm(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("M", C.c_int64), ("N", C.c_int), ("K", C.c_int),
                ("epilogue", C.c_int), ("bias", C.c_void_p), ("aux", C.c_void_p), ("out0", C.c_void_p),
                ("out1", C.c_void_p), ("colsum", C.c_void_p), ("drop", Dropout), ("embed", Embed)]


class Fp8Gemm(C.Structure):
    _fields_ = [("alpha", C.c_void_p), ("alpha2", C.c_void_p), ("out_fp8", C.c_void_p), ("out_scale", C.c_void_p),
                ("out_amax", C.c_void_p)]


EPI_BF16, EPI_F32, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_EMBED = range(6)

ABI_VERSION = 2          # vitssl_version(): 2 = vitssl_dino_loss takes the size of its scratch buffer
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype is always int except the two noted)
PROTOTYPES = {
    "vitssl_dropout_mask": [_vp, _i64, _i64, Dropout, _vp],
    "vitssl_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp],
    "vitssl_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, Dropout, _i64, _i, _vp],
    "vitssl_grad_mask_cast": [_vp, _vp, _vp, Dropout, _i64, _i, _vp],
    "vitssl_gemm_bf16_nt": [C.POINTER(Gemm), _vp],
    "vitssl_gemm_bf16_tn": [_vp, _vp, _vp, _i64, _i, _i, _vp, _i64, _vp],
    "vitssl_gemm_bf16_tn_batch": [C.POINTER(TnJob), _i, _i64, _vp, _i64, _vp],
    "vitssl_gemm_fp8_tn_batch": [C.POINTER(Fp8TnJob), _i, _i64, _vp, _i64, _vp],
    "vitssl_gemm_fp8_nt": [C.POINTER(Gemm), C.POINTER(Fp8Gemm), _vp],
    "vitssl_gemm_fp8_tn": [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _i64, _vp],
    "vitssl_quantize_fp8": [_vp, _vp, _i64, _vp],
    "vitssl_attn_bwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitssl_quantize_fp8_scaled": [_vp, _vp, _i64, _vp, _vp, _vp],
    "vitssl_layernorm_bwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, Dropout, _i64, _i, _vp],
    "vitssl_grad_mask_cast_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, Dropout, _i64, _i, _vp],
    "vitssl_attn_fwd_fp8": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitssl_layernorm_fwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp],
    "vitssl_fp8_quantize_weights": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "vitssl_attn_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitssl_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitssl_patchify_bf16": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vitssl_gather_patches_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vitssl_gather_rows_bf16": [_vp, _vp, _vp, _i, _i, _vp],
    "vitssl_scatter_rows_f32": [_vp, _vp, _vp, _i64, _i, _vp],
    "vitssl_gather_cls_f32": [_vp, _vp, _i, _i, _i, _vp],
    "vitssl_scatter_cls_f32": [_vp, _vp, _i, _i, _i, _vp],
    "vitssl_embed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp],
    "vitssl_l1_loss": [_vp, _vp, _vp, _vp, _f, _i64, _vp],
    "vitssl_cross_entropy": [_vp, _vp, _vp, _vp, _f, _i, _i, _vp],
    "vitssl_colsum_bf16": [_vp, _vp, _i64, _i, _vp],
    "vitssl_cast_bf16": [_vp, _vp, _i64, _vp],
    "vitssl_cast_transpose_bf16": [_vp, _vp, _vp, _i, _i, _vp],
    "vitssl_cast_transpose_batch": [_vp, _vp, _i, _i, _vp],
    "vitssl_bicubic_resize_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vitssl_bicubic_resize_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vitssl_aug_resized_crop_u8": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vitssl_aug_color_u8": [_vp, _vp, _vp, _i, _i, _vp],
    "vitssl_aug_blur_to_tensor": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "vitssl_adamw": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp],
    "vitssl_ema": [_vp, _vp, _i64, _f, _vp],
    "vitssl_rownorm_fwd": [_vp, _vp, _vp, _i64, _i, _vp],
    "vitssl_rownorm_bwd": [_vp, _vp, _vp, _vp, _i64, _i, _vp],
    "vitssl_weightnorm_fold": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "vitssl_weightnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "vitssl_dino_loss": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp],
    "vitssl_colsum_f32": [_vp, _vp, _i64, _i, _vp],
    "vitssl_center_ema": [_vp, _vp, _i, _f, _f, _vp],
    "vitssl_set_reserved_cus": [_i],
}

_lib = None


def header_symbols():
    """Entry points declared in include/vitssl_hip.h (int-returning `vitssl_*` functions)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(vitssl_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitsslError(
            f"{LIB_PATH} not found: the HIP library is not built. Run `python __graft_entry__.py` "
            "(or __graft_entry__.build()). There is no CPU fallback for the vit_core hot path.")
    # PyTorch-ROCm bundles its own libamdhip64; the process must end up with ONE HIP runtime
    # (device pointers and streams cross this boundary).  Loading torch first makes the
    # dynamic loader satisfy this library's libamdhip64.so.7 dependency with torch's copy;
    # the other order pulls in /opt/rocm's runtime as a second instance and every launch fails
    # with "no ROCm-capable device is detected" (seen with build() followed by smoke()).
    import torch  # noqa: F401
    l = C.CDLL(LIB_PATH)
    l.vitssl_last_error.restype = C.c_char_p
    l.vitssl_last_error.argtypes = []
    l.vitssl_version.restype = C.c_int
    l.vitssl_version.argtypes = []
    if l.vitssl_version() != ABI_VERSION:
        raise VitsslError(f"{LIB_PATH} implements C-ABI version {l.vitssl_version()}, these bindings expect {ABI_VERSION}: "
                          "rebuild it (python __graft_entry__.py)")
    l.vitssl_gemm_tn_workspace_floats.restype = C.c_int64
    l.vitssl_gemm_tn_workspace_floats.argtypes = [C.c_int64, C.c_int, C.c_int]
    l.vitssl_gemm_tn_batch_workspace_floats.restype = C.c_int64
    l.vitssl_gemm_tn_batch_workspace_floats.argtypes = [C.POINTER(TnJob), C.c_int, C.c_int64]
    l.vitssl_gemm_fp8_tn_batch_workspace_floats.restype = C.c_int64
    l.vitssl_gemm_fp8_tn_batch_workspace_floats.argtypes = [C.POINTER(Fp8TnJob), C.c_int, C.c_int64]
    l.vitssl_gemm_fp8_tn_workspace_floats.restype = C.c_int64
    l.vitssl_gemm_fp8_tn_workspace_floats.argtypes = [C.c_int64, C.c_int, C.c_int]
    l.vitssl_dino_loss_workspace_floats.restype = C.c_int64
    l.vitssl_dino_loss_workspace_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    l.vitssl_embed_bwd_workspace_floats.restype = C.c_int64
    l.vitssl_embed_bwd_workspace_floats.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    for getter in ("vitssl_get_reserved_cus", "vitssl_debug_last_nt_grid", "vitssl_debug_last_attn_fwd_grid"):
        getattr(l, getter).restype = C.c_int
        getattr(l, getter).argtypes = []
    for name, args in PROTOTYPES.items():
        fn = getattr(l, name)  # AttributeError if the symbol is missing
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = l
    return l


def call(name, *args):
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise VitsslError(f"{name} failed ({rc}): {l.vitssl_last_error().decode()}")
