"""Tensor-level wrappers over the C ABI: validate device/dtype/shape/contiguity on the
host (a wrong shape must never reach a hand-written kernel), then enqueue on torch's
current stream.  No computation happens in Python and there is no fallback path."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import Dropout, Embed, Fp8Gemm, Gemm, call

BF16 = torch.bfloat16
F32 = torch.float32
FP8 = torch.float8_e4m3fn      # OCP e4m3fn: what gfx950's conversion and MFMA instructions implement


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name, shape=None):
    if t is None:
        raise L.VitsslError(f"{name}: tensor is None")
    if not t.is_cuda:
        raise L.VitsslError(f"{name}: expected a CUDA (HIP) tensor, got {t.device}; there is no CPU fallback")
    if t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's current stream: a tensor of another GPU would be
        # touched by a kernel enqueued on the wrong device
        raise L.VitsslError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                            "call torch.cuda.set_device (one process per GPU) before using the engine")
    if t.dtype != dtype:
        raise L.VitsslError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.VitsslError(f"{name}: tensor must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise L.VitsslError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return C.c_void_p(t.data_ptr())


def _opt(t, dtype, name, shape=None):
    return C.c_void_p(0) if t is None else _chk(t, dtype, name, shape)


def make_dropout(p=0.0, seed=0, site=0):
    return Dropout(float(p), int(site) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFFFFFFFFFF)


NO_DROP = make_dropout()

# Optional live kernel timing (bench.py): when PROFILE is a list, every GEMM / attention launch and
# the HBM-bound kernels (LayerNorm, AdamW) are bracketed by HIP events ON THE LAUNCH STREAM and
# (label, flops, start, end, algorithmic_bytes) appended.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    return ev


def _prof_end(ev0, label, flops, nbytes=0):
    if ev0 is None:
        return
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record(torch.cuda.current_stream())
    PROFILE.append((label, flops, ev0, ev1, nbytes))


def dropout_mask(rows, cols, drop, device):
    keep = torch.empty(rows, cols, dtype=torch.uint8, device=device)
    call("vitssl_dropout_mask", _chk(keep, torch.uint8, "keep"), rows, cols, drop, _stream())
    return keep


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps=1e-5):
    rows, cols = x.shape
    ev = _prof_begin()
    call("vitssl_layernorm_fwd", _chk(x, F32, "x"), _chk(gamma, F32, "gamma", (cols,)), _chk(beta, F32, "beta", (cols,)),
         _chk(y, BF16, "y", (rows, cols)), _chk(mean, F32, "mean", (rows,)), _chk(rstd, F32, "rstd", (rows,)),
         rows, cols, float(eps), _stream())
    _prof_end(ev, "ln_fwd", 0.0, rows * (6 * cols + 8))          # x fp32 in, y bf16 out, mean / rstd


def layernorm_fwd_fp8(x, gamma, beta, y, y8, mean, rstd, eps=1e-5):
    """LayerNorm forward that writes the e4m3 image `y8` of its output (fp8 operand path); the bf16 image `y` is optional."""
    rows, cols = x.shape
    ev = _prof_begin()
    call("vitssl_layernorm_fwd_fp8", _chk(x, F32, "x"), _chk(gamma, F32, "gamma", (cols,)), _chk(beta, F32, "beta", (cols,)),
         _opt(y, BF16, "y", (rows, cols)), _chk(y8, FP8, "y8", (rows, cols)), _chk(mean, F32, "mean", (rows,)),
         _chk(rstd, F32, "rstd", (rows,)), rows, cols, float(eps), _stream())
    _prof_end(ev, "ln_fwd", 0.0, rows * ((7 if y is not None else 5) * cols + 8))


def _scalar(t, name):
    if t is None:
        return C.c_void_p(0)
    if t.numel() != 1:
        raise L.VitsslError(f"{name}: expected a 1-element fp32 device tensor")
    return _chk(t, F32, name)


def gemm_fp8_tn_batch(jobs):
    """For every (A8, B8, Cacc, alpha, alpha2) of `jobs` (at most 8, same row count M): Cacc += alpha * alpha2 * A8^T @ B8 on e4m3
    operands, in one launch (vitssl_gemm_fp8_tn_batch)."""
    if not jobs:
        return
    M = jobs[0][0].shape[0]
    arr = (L.Fp8TnJob * len(jobs))()
    flops = 0.0
    for j, (A8, B8, Cacc, alpha, alpha2) in enumerate(jobs):
        if A8.shape[0] != M or B8.shape[0] != M:
            raise L.VitsslError(f"gemm_fp8_tn_batch: job {j}: row counts {A8.shape[0]} / {B8.shape[0]} differ from {M}")
        N1, N2 = A8.shape[1], B8.shape[1]
        a, b, c = _chk(A8, FP8, "A8"), _chk(B8, FP8, "B8"), _chk(Cacc, F32, "C", (N1, N2))
        arr[j].A8, arr[j].B8, arr[j].C, arr[j].N1, arr[j].N2 = a.value, b.value, c.value, N1, N2
        arr[j].alpha, arr[j].alpha2 = _scalar(alpha, "alpha").value, _scalar(alpha2, "alpha2").value
        flops += 2.0 * M * N1 * N2
    wsn = int(L.lib().vitssl_gemm_fp8_tn_batch_workspace_floats(arr, len(jobs), M))
    ws = _tn_workspace(jobs[0][0].device, wsn)
    ev = _prof_begin()
    call("vitssl_gemm_fp8_tn_batch", arr, len(jobs), M, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _prof_end(ev, f"gemm_fp8_tn batch{len(jobs)}x{M}", flops)


def quantize_fp8(x, y8, scale=None, amax=None):
    """y8 = e4m3(clamp(x * scale, +-448)) of a bf16 tensor; `scale` / `amax` are optional 1-element device tensors
    (amax receives max(amax, max|x|))."""
    call("vitssl_quantize_fp8_scaled", _chk(x, BF16, "x"), _chk(y8, FP8, "y8", x.shape), x.numel(), _scalar(scale, "scale"),
         _scalar(amax, "amax"), _stream())


def layernorm_bwd_fp8(dy, x, mean, rstd, gamma, g_res, g_out, gm, gm8, scale, amax, dgamma, dbeta, gm_colsum=None, drop=NO_DROP):
    """layernorm_bwd that also writes gm8 = e4m3(gm * scale) and records max|gm| in amax."""
    rows, cols = x.shape
    ev = _prof_begin()
    call("vitssl_layernorm_bwd_fp8", _chk(dy, BF16, "dy", (rows, cols)), _chk(x, F32, "x"), _chk(mean, F32, "mean", (rows,)),
         _chk(rstd, F32, "rstd", (rows,)), _chk(gamma, F32, "gamma", (cols,)), _opt(g_res, F32, "g_res", (rows, cols)),
         _chk(g_out, F32, "g_out", (rows, cols)), _opt(gm, BF16, "gm", (rows, cols)), _chk(gm8, FP8, "gm8", (rows, cols)),
         _scalar(scale, "scale"), _scalar(amax, "amax"), _chk(dgamma, F32, "dgamma", (cols,)), _chk(dbeta, F32, "dbeta", (cols,)),
         _opt(gm_colsum, F32, "gm_colsum", (cols,)), drop, rows, cols, _stream())
    _prof_end(ev, "ln_bwd", 0.0, rows * (cols * (2 + 4 + (4 if g_res is not None else 0) + 4 + (3 if gm is not None else 1)) + 8))


def grad_mask_cast_fp8(g, gm, gm8, scale, amax, gm_colsum=None, drop=NO_DROP):
    rows, cols = g.shape
    call("vitssl_grad_mask_cast_fp8", _chk(g, F32, "g"), _opt(gm, BF16, "gm", (rows, cols)), _chk(gm8, FP8, "gm8", (rows, cols)),
         _scalar(scale, "scale"), _scalar(amax, "amax"), _opt(gm_colsum, F32, "gm_colsum", (cols,)), drop, rows, cols, _stream())


def layernorm_bwd(dy, x, mean, rstd, gamma, g_res, g_out, gm, dgamma, dbeta, gm_colsum=None, drop=NO_DROP):
    rows, cols = x.shape
    ev = _prof_begin()
    call("vitssl_layernorm_bwd", _chk(dy, BF16, "dy", (rows, cols)), _chk(x, F32, "x"), _chk(mean, F32, "mean", (rows,)),
         _chk(rstd, F32, "rstd", (rows,)), _chk(gamma, F32, "gamma", (cols,)), _opt(g_res, F32, "g_res", (rows, cols)),
         _chk(g_out, F32, "g_out", (rows, cols)), _opt(gm, BF16, "gm", (rows, cols)), _chk(dgamma, F32, "dgamma", (cols,)),
         _chk(dbeta, F32, "dbeta", (cols,)), _opt(gm_colsum, F32, "gm_colsum", (cols,)), drop, rows, cols, _stream())
    # dy bf16 + x fp32 (+ residual gradient fp32) in, gradient fp32 (+ masked bf16 operand) out
    _prof_end(ev, "ln_bwd", 0.0, rows * (cols * (2 + 4 + (4 if g_res is not None else 0) + 4 + (2 if gm is not None else 0)) + 8))


def grad_mask_cast(g, gm, gm_colsum=None, drop=NO_DROP):
    rows, cols = g.shape
    call("vitssl_grad_mask_cast", _chk(g, F32, "g"), _chk(gm, BF16, "gm", (rows, cols)),
         _opt(gm_colsum, F32, "gm_colsum", (cols,)), drop, rows, cols, _stream())


_OUT0_DTYPE = {L.EPI_BF16: BF16, L.EPI_F32: F32, L.EPI_GELU: BF16, L.EPI_RESID: F32, L.EPI_DGELU: BF16, L.EPI_EMBED: F32}


def gemm_nt(A, B, out0, epilogue, bias=None, aux=None, out1=None, colsum=None, drop=NO_DROP, embed=None):
    """out = A[M,K] @ B[N,K]^T with the fused epilogue (see include/vitssl_hip.h)."""
    M, K = A.shape
    N, K2 = B.shape
    if K != K2:
        raise L.VitsslError(f"gemm_nt: K mismatch {K} vs {K2}")
    g = Gemm()
    g.A = _chk(A, BF16, "A")
    g.B = _chk(B, BF16, "B")
    g.M, g.N, g.K = M, N, K
    g.epilogue = epilogue
    g.bias = _opt(bias, F32, "bias", (N,))
    if epilogue == L.EPI_EMBED:
        if embed is None:
            raise L.VitsslError("gemm_nt: EPI_EMBED needs embed=(mask, mask_token, pos, tokens, out_tokens, tok_offset)")
        mask, mask_token, pos, tokens, out_tokens, tok_offset = embed
        if M % tokens != 0:
            raise L.VitsslError("gemm_nt: M must be a multiple of tokens")
        e = Embed()
        e.mask = _opt(mask, torch.uint8, "mask", (M,))
        e.mask_token = _opt(mask_token, F32, "mask_token", (N,))
        e.pos = _chk(pos, F32, "pos", (out_tokens, N))
        e.tokens, e.out_tokens, e.tok_offset = tokens, out_tokens, tok_offset
        g.embed = e
        g.out0 = _chk(out0, F32, "out0", ((M // tokens) * out_tokens, N))
    else:
        g.out0 = _chk(out0, _OUT0_DTYPE[epilogue], "out0", (M, N))
    if epilogue == L.EPI_RESID:
        g.aux = _chk(aux, F32, "aux(residual)", (M, N))
    elif epilogue == L.EPI_DGELU:
        g.aux = _chk(aux, BF16, "aux(u)", (M, N))
    g.out1 = _opt(out1, BF16, "out1", (M, N))
    g.colsum = _opt(colsum, F32, "colsum", (N,))
    g.drop = drop
    ev = _prof_begin()
    call("vitssl_gemm_bf16_nt", C.byref(g), _stream())
    _prof_end(ev, f"gemm_nt[epi{epilogue}] {M}x{N}x{K}", 2.0 * M * N * K)


def gemm_fp8_nt(A8, B8, out0, epilogue, alpha=None, bias=None, aux=None, out1=None, out_fp8=None, drop=NO_DROP,
                alpha2=None, colsum=None, out_scale=None, out_amax=None):
    """out = alpha * alpha2 * (A8[M,K] @ B8[N,K]^T) with the fused epilogue; A8 / B8 are e4m3 operands, `alpha` /
    `alpha2` 1-element device tensors (dequantisation factors of the two operands).  `out_fp8`: optional e4m3 image of
    out1 (EPI_GELU) or of out0 (EPI_DGELU), written as e4m3(value * out_scale) with max|value| recorded in out_amax.
    Forward and input-gradient GEMMs (include/vitssl_hip.h)."""
    M, K = A8.shape
    N, K2 = B8.shape
    if K != K2:
        raise L.VitsslError(f"gemm_fp8_nt: K mismatch {K} vs {K2}")
    if epilogue not in (L.EPI_BF16, L.EPI_F32, L.EPI_GELU, L.EPI_RESID, L.EPI_DGELU):
        raise L.VitsslError(f"gemm_fp8_nt: epilogue {epilogue} has no fp8-operand form")
    g = Gemm()
    g.A = _chk(A8, FP8, "A8")
    g.B = _chk(B8, FP8, "B8")
    g.M, g.N, g.K = M, N, K
    g.epilogue = epilogue
    g.bias = _opt(bias, F32, "bias", (N,))
    if out0 is None and not (epilogue == L.EPI_DGELU and out_fp8 is not None):
        raise L.VitsslError("gemm_fp8_nt: out0 may only be omitted for EPI_DGELU with out_fp8")
    g.out0 = _opt(out0, _OUT0_DTYPE[epilogue], "out0", (M, N))
    if epilogue == L.EPI_RESID:
        g.aux = _chk(aux, F32, "aux(residual)", (M, N))
    elif epilogue == L.EPI_DGELU:
        g.aux = _chk(aux, BF16, "aux(g')", (M, N))
    g.out1 = _opt(out1, BF16, "out1", (M, N))
    g.colsum = _opt(colsum, F32, "colsum", (N,))
    g.drop = drop
    q = Fp8Gemm()
    q.alpha = _scalar(alpha, "alpha")
    q.alpha2 = _scalar(alpha2, "alpha2")
    q.out_fp8 = _opt(out_fp8, FP8, "out_fp8", (M, N))
    q.out_scale = _scalar(out_scale, "out_scale")
    q.out_amax = _scalar(out_amax, "out_amax")
    if out_fp8 is not None and epilogue not in (L.EPI_GELU, L.EPI_DGELU):
        raise L.VitsslError("gemm_fp8_nt: out_fp8 belongs to EPI_GELU / EPI_DGELU")
    ev = _prof_begin()
    call("vitssl_gemm_fp8_nt", C.byref(g), C.byref(q), _stream())
    _prof_end(ev, f"gemm_fp8_nt[epi{epilogue}] {M}x{N}x{K}", 2.0 * M * N * K)


_TN_WS = {}


def _tn_workspace(device, floats):
    """Per-device slab workspace for the split-M wgrad combine (grown on demand, reused by
    every launch: launches on one stream are ordered, the reduce kernel drains the slabs
    before the next wgrad writes them)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 20), dtype=F32, device=device)
        _TN_WS[key] = ws
    return ws


def gemm_tn(A, B, Cacc, atomic=False):
    """Cacc[N1,N2] (fp32) += A[M,N1]^T @ B[M,N2]."""
    M, N1 = A.shape
    M2, N2 = B.shape
    if M != M2:
        raise L.VitsslError(f"gemm_tn: M mismatch {M} vs {M2}")
    a, b, c = _chk(A, BF16, "A"), _chk(B, BF16, "B"), _chk(Cacc, F32, "C", (N1, N2))
    if atomic:
        wsp, wsn = C.c_void_p(0), 0
    else:
        wsn = int(L.lib().vitssl_gemm_tn_workspace_floats(M, N1, N2))
        ws = _tn_workspace(A.device, wsn)
        wsp, wsn = C.c_void_p(ws.data_ptr()), ws.numel()
    ev = _prof_begin()
    call("vitssl_gemm_bf16_tn", a, b, c, M, N1, N2, wsp, wsn, _stream())
    _prof_end(ev, f"gemm_tn {N1}x{N2}x{M}", 2.0 * M * N1 * N2)


def gemm_tn_batch(jobs):
    """For every (A, B, Cacc) of `jobs` (at most 8, all with the same row count M): Cacc[N1,N2] (fp32) += A[M,N1]^T @ B[M,N2], in
    one launch with one shared split count and one reduce pass (vitssl_gemm_bf16_tn_batch)."""
    if not jobs:
        return
    M = jobs[0][0].shape[0]
    arr = (L.TnJob * len(jobs))()
    flops = 0.0
    for j, (A, B, Cacc) in enumerate(jobs):
        if A.shape[0] != M or B.shape[0] != M:
            raise L.VitsslError(f"gemm_tn_batch: job {j}: row counts {A.shape[0]} / {B.shape[0]} differ from {M}")
        N1, N2 = A.shape[1], B.shape[1]
        a, b, c = _chk(A, BF16, "A"), _chk(B, BF16, "B"), _chk(Cacc, F32, "C", (N1, N2))
        arr[j].A, arr[j].B, arr[j].C, arr[j].N1, arr[j].N2 = a.value, b.value, c.value, N1, N2
        flops += 2.0 * M * N1 * N2
    wsn = int(L.lib().vitssl_gemm_tn_batch_workspace_floats(arr, len(jobs), M))
    ws = _tn_workspace(jobs[0][0].device, wsn)
    ev = _prof_begin()
    call("vitssl_gemm_bf16_tn_batch", arr, len(jobs), M, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _prof_end(ev, f"gemm_tn batch{len(jobs)}x{M}", flops)


def gemm_fp8_tn(A8, B8, Cacc, alpha=None, alpha2=None):
    """Cacc[N1,N2] (fp32) += alpha * alpha2 * A8[M,N1]^T @ B8[M,N2] on e4m3 operands (weight gradient of the fp8 path)."""
    M, N1 = A8.shape
    M2, N2 = B8.shape
    if M != M2:
        raise L.VitsslError(f"gemm_fp8_tn: M mismatch {M} vs {M2}")
    a, b, c = _chk(A8, FP8, "A8"), _chk(B8, FP8, "B8"), _chk(Cacc, F32, "C", (N1, N2))
    wsn = int(L.lib().vitssl_gemm_fp8_tn_workspace_floats(M, N1, N2))
    ws = _tn_workspace(A8.device, wsn)
    ev = _prof_begin()
    call("vitssl_gemm_fp8_tn", a, b, c, M, N1, N2, _scalar(alpha, "alpha"), _scalar(alpha2, "alpha2"), C.c_void_p(ws.data_ptr()),
         ws.numel(), _stream())
    _prof_end(ev, f"gemm_fp8_tn {N1}x{N2}x{M}", 2.0 * M * N1 * N2)


def attn_fwd(qkv, out, lse, B, N, H, dh, probs=None, out_fp8=None):
    ev = _prof_begin()
    _attn_fwd(qkv, out, lse, B, N, H, dh, probs, out_fp8)
    _prof_end(ev, f"attn_fwd B{B} N{N} H{H}", 4.0 * B * H * N * N * dh)


def _attn_fwd(qkv, out, lse, B, N, H, dh, probs=None, out_fp8=None):
    if out_fp8 is not None:       # fp8 operand path: also the e4m3 image of `out`
        call("vitssl_attn_fwd_fp8", _chk(qkv, BF16, "qkv", (B * N, 3 * H * dh)), _chk(out, BF16, "out", (B * N, H * dh)),
             _chk(out_fp8, FP8, "out_fp8", (B * N, H * dh)), _chk(lse, F32, "lse", (B, H, N)),
             _opt(probs, F32, "probs", (B, H, N, N)), B, N, H, dh, _stream())
        return
    call("vitssl_attn_fwd", _chk(qkv, BF16, "qkv", (B * N, 3 * H * dh)), _chk(out, BF16, "out", (B * N, H * dh)),
         _chk(lse, F32, "lse", (B, H, N)), _opt(probs, F32, "probs", (B, H, N, N)), B, N, H, dh, _stream())


def attn_bwd(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, dh, dqkv_fp8=None, scale=None, amax=None):
    """`dqkv_fp8` (fp8 path): also the e4m3 image e4m3(dqkv * scale), with max|dqkv| recorded in `amax`."""
    ev = _prof_begin()
    if dqkv_fp8 is not None:
        call("vitssl_attn_bwd_fp8", _chk(qkv, BF16, "qkv", (B * N, 3 * H * dh)), _chk(out, BF16, "out", (B * N, H * dh)),
             _chk(dout, BF16, "dout", (B * N, H * dh)), _chk(lse, F32, "lse", (B, H, N)), _opt(dqkv, BF16, "dqkv", (B * N, 3 * H * dh)),
             _chk(dqkv_fp8, FP8, "dqkv_fp8", (B * N, 3 * H * dh)), _scalar(scale, "scale"), _scalar(amax, "amax"), B, N, H, dh, _stream())
    else:
        _attn_bwd(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, dh)
    _prof_end(ev, f"attn_bwd B{B} N{N} H{H}", 8.0 * B * H * N * N * dh)


def _attn_bwd(qkv, out, dout, lse, dqkv, delta_ws, B, N, H, dh):
    call("vitssl_attn_bwd", _chk(qkv, BF16, "qkv", (B * N, 3 * H * dh)), _chk(out, BF16, "out", (B * N, H * dh)),
         _chk(dout, BF16, "dout", (B * N, H * dh)), _chk(lse, F32, "lse", (B, H, N)),
         _chk(dqkv, BF16, "dqkv", (B * N, 3 * H * dh)), _chk(delta_ws, F32, "delta_ws", (B, H, N)), B, N, H, dh, _stream())


def patchify_bf16(img, patches, P):
    B, Cc, H, W = img.shape
    call("vitssl_patchify_bf16", _chk(img, F32, "img"), _chk(patches, BF16, "patches", (B * (H // P) * (W // P), Cc * P * P)),
         B, Cc, H, W, P, _stream())


def gather_patches_f32(img, idx, out, P):
    B, Cc, H, W = img.shape
    n = idx.numel()
    call("vitssl_gather_patches_f32", _chk(img, F32, "img"), _chk(idx, torch.int32, "idx"), _chk(out, F32, "out", (n, Cc * P * P)),
         n, Cc, H, W, P, _stream())


def gather_rows_bf16(x, idx, out):
    rows, cols = x.shape
    n = idx.numel()
    call("vitssl_gather_rows_bf16", _chk(x, F32, "x"), _chk(idx, torch.int32, "idx"), _chk(out, BF16, "out", (n, cols)), n, cols, _stream())


def scatter_rows_f32(src, inv, g):
    rows, cols = g.shape
    call("vitssl_scatter_rows_f32", _chk(src, BF16, "src"), _chk(inv, torch.int32, "inv", (rows,)), _chk(g, F32, "g"), rows, cols, _stream())


def gather_cls_f32(x, out, B, T, D):
    call("vitssl_gather_cls_f32", _chk(x, F32, "x", (B * T, D)), _chk(out, F32, "out", (B, D)), B, T, D, _stream())


def scatter_cls_f32(gcls, g, B, T, D):
    call("vitssl_scatter_cls_f32", _chk(gcls, F32, "gcls", (B, D)), _chk(g, F32, "g", (B * T, D)), B, T, D, _stream())


def embed_bwd(dtok, mask, dproj, dpos, dmask_token, dbias, dcls, B, tokens, tok_offset, D):
    T_out = tokens + tok_offset
    wsn = int(L.lib().vitssl_embed_bwd_workspace_floats(B, tokens, tok_offset, D))
    ws = _tn_workspace(dtok.device, wsn)       # shared scratch: launches on one stream are ordered
    call("vitssl_embed_bwd", _chk(dtok, F32, "dtok", (B * T_out, D)), _opt(mask, torch.uint8, "mask", (B * tokens,)),
         _chk(dproj, BF16, "dproj", (B * tokens, D)), _opt(dpos, F32, "dpos", (T_out, D)),
         _opt(dmask_token, F32, "dmask_token", (D,)), _opt(dbias, F32, "dbias", (D,)), _opt(dcls, F32, "dcls", (D,)),
         B, tokens, tok_offset, D, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())


def l1_loss(pred, target, loss_sum, dpred=None, gscale=0.0):
    n = pred.numel()
    call("vitssl_l1_loss", _chk(pred, F32, "pred"), _chk(target, F32, "target", pred.shape), _chk(loss_sum, F32, "loss_sum", (1,)),
         _opt(dpred, BF16, "dpred", pred.shape), float(gscale), n, _stream())


def cross_entropy(logits, labels, loss_sum, dlogits=None, gscale=0.0):
    B, Cn = logits.shape
    call("vitssl_cross_entropy", _chk(logits, F32, "logits"), _chk(labels, torch.int64, "labels", (B,)),
         _chk(loss_sum, F32, "loss_sum", (1,)), _opt(dlogits, BF16, "dlogits", (B, Cn)), float(gscale), B, Cn, _stream())


def colsum_bf16(x, out):
    rows, cols = x.shape
    call("vitssl_colsum_bf16", _chk(x, BF16, "x"), _chk(out, F32, "out", (cols,)), rows, cols, _stream())


def cast_bf16(src, dst):
    call("vitssl_cast_bf16", _chk(src, F32, "src"), _chk(dst, BF16, "dst", src.shape), src.numel(), _stream())


def cast_transpose_bf16(src, dst, dst_t):
    R, Cn = src.shape
    call("vitssl_cast_transpose_bf16", _chk(src, F32, "src"), _opt(dst, BF16, "dst", (R, Cn)), _opt(dst_t, BF16, "dst_t", (Cn, R)),
         R, Cn, _stream())


class CastPlan:
    """Device-resident job table for `cast_transpose_batch`: built once per set of
    (source, destination) pointers, re-uploaded only when a pointer or shape changes."""

    def __init__(self):
        self.key = None
        self.jobs_dev = None
        self.starts_dev = None
        self.njobs = 0
        self.total = 0
        self.keep = None

    def run(self, jobs):
        """jobs: list of (src f32 [R,C], dst bf16 [R,C] | None, dst_t bf16 [C,R] | None)."""
        import numpy as np
        key = tuple((s.data_ptr(), 0 if d is None else d.data_ptr(), 0 if t is None else t.data_ptr(), s.shape[0], s.shape[1])
                    for s, d, t in jobs)
        if key != self.key:
            for s, d, t in jobs:
                R, Cn = s.shape
                _chk(s, F32, "src"); _opt(d, BF16, "dst", (R, Cn)); _opt(t, BF16, "dst_t", (Cn, R))
            dev = jobs[0][0].device
            rec = np.zeros(len(jobs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("dst_t", "<u8"), ("R", "<i4"), ("C", "<i4")]))
            starts = np.zeros(len(jobs) + 1, dtype=np.int32)
            for i, k in enumerate(key):
                rec[i] = k
                starts[i + 1] = starts[i] + ((k[3] + 63) // 64) * ((k[4] + 63) // 64)
            self.jobs_dev = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
            self.starts_dev = torch.from_numpy(starts).to(dev)
            self.njobs, self.total, self.key = len(jobs), int(starts[-1]), key
        self.keep = jobs        # the sources must outlive the launch
        call("vitssl_cast_transpose_batch", C.c_void_p(self.jobs_dev.data_ptr()), C.c_void_p(self.starts_dev.data_ptr()),
             self.njobs, self.total, _stream())


class Fp8WeightPlan:
    """Device-resident job table for `vitssl_fp8_quantize_weights` (per-tensor power-of-two scales computed on the
    device, no host synchronisation): `alpha` holds one dequantisation factor per job."""

    def __init__(self):
        self.key = None
        self.jobs_dev = self.starts_dev = self.amax = self.alpha = None
        self.njobs = self.total = 0
        self.keep = None

    def run(self, jobs):
        """jobs: list of (src f32 [R,C], dst e4m3 [R,C] | None, dst_t e4m3 [C,R] | None)."""
        import numpy as np
        key = tuple((s.data_ptr(), 0 if d is None else d.data_ptr(), 0 if t is None else t.data_ptr(), s.shape[0], s.shape[1])
                    for s, d, t in jobs)
        if key != self.key:
            for s, d, t in jobs:
                R, Cn = s.shape
                _chk(s, F32, "src"); _opt(d, FP8, "dst", (R, Cn)); _opt(t, FP8, "dst_t", (Cn, R))
            dev = jobs[0][0].device
            rec = np.zeros(len(jobs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("dst_t", "<u8"), ("R", "<i4"), ("C", "<i4")]))
            starts = np.zeros(len(jobs) + 1, dtype=np.int32)
            for i, k in enumerate(key):
                rec[i] = k
                starts[i + 1] = starts[i] + ((k[3] + 63) // 64) * ((k[4] + 63) // 64)
            self.jobs_dev = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
            self.starts_dev = torch.from_numpy(starts).to(dev)
            self.amax = torch.zeros(len(jobs), dtype=F32, device=dev)
            self.alpha = torch.ones(len(jobs), dtype=F32, device=dev)
            self.njobs, self.total, self.key = len(jobs), int(starts[-1]), key
        self.keep = jobs
        call("vitssl_fp8_quantize_weights", C.c_void_p(self.jobs_dev.data_ptr()), C.c_void_p(self.starts_dev.data_ptr()),
             self.njobs, self.total, C.c_void_p(self.amax.data_ptr()), C.c_void_p(self.alpha.data_ptr()), _stream())


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, gscale=1.0):
    n = p.numel()
    ev = _prof_begin()
    call("vitssl_adamw", _chk(p, F32, "p"), _chk(g, F32, "g", p.shape), _chk(m, F32, "m", p.shape), _chk(v, F32, "v", p.shape),
         n, float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), float(gscale), _stream())
    _prof_end(ev, "adamw", 0.0, 28 * n)                          # p, m, v read + written, g read


def ema(teacher, student, m):
    ev = _prof_begin()
    call("vitssl_ema", _chk(teacher, F32, "teacher"), _chk(student, F32, "student", teacher.shape), teacher.numel(), float(m), _stream())
    _prof_end(ev, "ema", 0.0, 12 * teacher.numel())               # teacher read + written, student read


# ---- DINO ----------------------------------------------------------------------------
def rownorm_fwd(z, zn, inv_norm):
    rows, cols = z.shape
    call("vitssl_rownorm_fwd", _chk(z, F32, "z"), _chk(zn, BF16, "zn", (rows, cols)), _chk(inv_norm, F32, "inv_norm", (rows,)), rows, cols, _stream())


def rownorm_bwd(dzn, zn, inv_norm, dz):
    rows, cols = dzn.shape
    call("vitssl_rownorm_bwd", _chk(dzn, F32, "dzn"), _chk(zn, BF16, "zn", (rows, cols)), _chk(inv_norm, F32, "inv_norm", (rows,)),
         _chk(dz, BF16, "dz", (rows, cols)), rows, cols, _stream())


def weightnorm_fold(g, v, w_f32, inv_vnorm):
    K, D = v.shape
    call("vitssl_weightnorm_fold", _chk(g, F32, "g"), _chk(v, F32, "v"), _chk(w_f32, F32, "w", (K, D)), _chk(inv_vnorm, F32, "inv_vnorm", (K,)),
         K, D, _stream())


def weightnorm_bwd(dw, g, v, inv_vnorm, dg, dv):
    K, D = v.shape
    call("vitssl_weightnorm_bwd", _chk(dw, F32, "dw", (K, D)), _chk(g, F32, "g"), _chk(v, F32, "v"), _chk(inv_vnorm, F32, "inv_vnorm", (K,)),
         _chk(dg, F32, "dg"), _chk(dv, F32, "dv", (K, D)), K, D, _stream())


def dino_tws_floats(G, B, K):
    """size of vitssl_dino_loss's scratch: [B, K] teacher probabilities + VITSSL_DINO_TWS_EXTRA(G, B) statistics"""
    return int(L.lib().vitssl_dino_loss_workspace_floats(G, B, K))


def dino_loss(teacher, student, center, t_ws, loss_sum, dstudent, G, V, B, K, teacher_temp, student_temp, gscale=1.0):
    ev = _prof_begin()
    _dino_loss(teacher, student, center, t_ws, loss_sum, dstudent, G, V, B, K, teacher_temp, student_temp, gscale)
    # teacher / student logits read (fp32), teacher-probability scratch written + read, bf16 gradient written
    _prof_end(ev, "dino_loss", 0.0, B * K * (4 * (G + V) + 8 + (2 * V if dstudent is not None else 0)))


def _dino_loss(teacher, student, center, t_ws, loss_sum, dstudent, G, V, B, K, teacher_temp, student_temp, gscale=1.0):
    call("vitssl_dino_loss", _chk(teacher, F32, "teacher", (G * B, K)), _chk(student, F32, "student", (V * B, K)),
         _chk(center, F32, "center"), _chk(t_ws, F32, "t_ws"), t_ws.numel(), _chk(loss_sum, F32, "loss_sum", (1,)),
         _opt(dstudent, BF16, "dstudent", (V * B, K)), G, V, B, K, float(teacher_temp), float(student_temp), float(gscale), _stream())


def colsum_f32(x, out):
    rows, cols = x.shape
    ev = _prof_begin()
    call("vitssl_colsum_f32", _chk(x, F32, "x"), _chk(out, F32, "out", (cols,)), rows, cols, _stream())
    _prof_end(ev, "center_colsum", 0.0, 4 * rows * cols)


def center_ema(center, colsum, momentum, inv_rows):
    K = colsum.numel()
    call("vitssl_center_ema", _chk(center, F32, "center"), _chk(colsum, F32, "colsum"), K, float(momentum), float(inv_rows), _stream())


def bicubic_resize_fwd(src, dst, gh0, gw0, gh, gw):
    D = src.shape[1]
    call("vitssl_bicubic_resize_fwd", _chk(src, F32, "src", (gh0 * gw0, D)), _chk(dst, F32, "dst", (gh * gw, D)),
         gh0, gw0, gh, gw, D, _stream())


def bicubic_resize_bwd(ddst, dsrc, gh0, gw0, gh, gw):
    D = ddst.shape[1]
    call("vitssl_bicubic_resize_bwd", _chk(ddst, F32, "ddst", (gh * gw, D)), _chk(dsrc, F32, "dsrc", (gh0 * gw0, D)),
         gh0, gw0, gh, gw, D, _stream())


# ---- DINO multi-crop input pipeline (data/datasets.py:80-123) -------------------------------
AUG_IP, AUG_FP = 11, 10


def aug_resized_crop_u8(src, iparams, tmp, dst):
    B, H, W, Cc = src.shape
    S = dst.shape[1]
    if Cc != 3:
        raise L.VitsslError(f"aug_resized_crop: expected channel-last RGB [B,H,W,3], got {tuple(src.shape)}")
    call("vitssl_aug_resized_crop_u8", _chk(src, torch.uint8, "src"), _chk(iparams, torch.int32, "iparams", (B, AUG_IP)),
         _chk(tmp, torch.uint8, "tmp", (B, H, S, 3)), _chk(dst, torch.uint8, "dst", (B, S, S, 3)), B, H, W, S, _stream())


def aug_color_u8(img, iparams, fparams):
    B, S = img.shape[0], img.shape[1]
    call("vitssl_aug_color_u8", _chk(img, torch.uint8, "img", (B, S, S, 3)), _chk(iparams, torch.int32, "iparams", (B, AUG_IP)),
         _chk(fparams, F32, "fparams", (B, AUG_FP)), B, S, _stream())


def aug_blur_to_tensor(img, fparams, out, ksize=7):
    B, S = img.shape[0], img.shape[1]
    call("vitssl_aug_blur_to_tensor", _chk(img, torch.uint8, "img", (B, S, S, 3)), _chk(fparams, F32, "fparams", (B, AUG_FP)),
         _chk(out, F32, "out", (B, 3, S, S)), B, S, ksize, _stream())
