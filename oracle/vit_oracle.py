"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the reference hot path.

Independent, functional (state_dict in, tensors out) restatement of
kristi700/ViT-SSL ``vit_core`` forward math, written from the reference's
behaviour (file:line cited per function, paths relative to /root/reference).
Gradients come from torch autograd over these functions on CPU.

Two modes:
  * ``emu=None``  : pure fp32 -- this is the "reference PyTorch-CPU path"
    (on a CPU host the reference's autocast/GradScaler disable themselves,
    SURVEY.md section 8a "Autocast dtype contract").
  * ``emu="bf16"``: same math, but values are rounded to bf16 at exactly the
    points where the HIP path stores bf16 (GEMM operands, LN output, QKV,
    attention probabilities/context, MLP hidden).  Used by the GPU parity
    tests to separate "kernel bug" from "bf16 rounding" with a tight tolerance.

    Inside ``with flash_delta():`` the attention BACKWARD of this mode is the flash-style
    form the HIP path uses (delta = rowsum(dO * O) from the bf16 output; see sdpa()).

  * ``emu="fp8"`` : as "bf16", but the four Linear layers of every encoder block
    run their FORWARD product on OCP e4m3fn operands (BASELINE configs[4], "fp8
    weight path"): activations quantised at unit scale (saturating at +-448),
    weights per tensor with the power-of-two scale 2^floor(log2(448 / max|w|)),
    fp32 accumulation.  With the gradient scales the HIP engine exports
    (``fp8_gscales``) the backward products dX = dY.W and dW = dY^T.X run on e4m3
    operands as well (dY quantised with that per-tensor scale).  The reference has no fp8 code: this mode restates the HIP design,
    and is tied to the reference only through its fp32 / bf16 siblings.

Pinned against fixtures generated from the reference (tests/golden).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------
# rounding emulation
# --------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    """Round-to-nearest-even to bf16, straight-through gradient (the HIP
    backward treats stored bf16 activations as the exact forward values)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def rnd(x: Tensor, emu: Optional[str]) -> Tensor:
    if emu is None:
        return x
    if emu in ("bf16", "fp8"):
        return _RoundBF16.apply(x)
    raise ValueError(emu)


E4M3_MAX = 448.0


def q8(x: Tensor) -> Tensor:
    """fp32 -> OCP e4m3fn -> fp32: round to nearest even, saturating at +-448."""
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def fp8_scale_exp(amax: float) -> int:
    """k = floor(log2(448 / amax)) computed on the binary representation (no log2 rounding): amax = fr 2^e with
    fr in [0.5, 1), 448 = 0.875 2^9; k = 0 for amax = 0."""
    if not (amax > 0.0) or not math.isfinite(amax):
        return 0
    fr, e = math.frexp(float(torch.tensor(amax, dtype=torch.float32)))
    k = 9 - e - (0 if fr <= 0.875 else 1)
    return max(-120, min(120, k))


class _LinearFP8(torch.autograd.Function):
    """y = (q8(x) . q8(w 2^k)^T) 2^-k in the forward.  Backward without a gradient scale: bf16 operands (g . bf16(w),
    g^T . bf16(x)).  With a gradient scale s (the HIP fp8 path): both products on e4m3 operands,
    dX = q8(g s) . q8(w 2^k) 2^-k / s and dW = q8(g s)^T . q8(x) / s."""

    @staticmethod
    def forward(ctx, x32, w, k, via_bf16, gscale):
        xb = x32.to(torch.bfloat16).to(torch.float32)
        wb = w.to(torch.bfloat16).to(torch.float32)
        x8 = q8(xb if via_bf16 else x32)
        w8 = q8(w * (2.0 ** k))
        ctx.save_for_backward(xb, wb, w8, x8)
        ctx.k, ctx.gscale = k, gscale
        return (x8 @ w8.t()) * (2.0 ** -k)

    @staticmethod
    def backward(ctx, g):
        xb, wb, w8, x8 = ctx.saved_tensors
        if ctx.gscale is None:
            gx = g @ wb
            gw = g.reshape(-1, g.shape[-1]).t() @ xb.reshape(-1, xb.shape[-1])
        else:
            g8 = q8(g * ctx.gscale)
            gx = (g8 @ w8) * ((2.0 ** -ctx.k) / ctx.gscale)
            gw = (g8.reshape(-1, g.shape[-1]).t() @ x8.reshape(-1, x8.shape[-1])) / ctx.gscale
        return gx, gw, None, None, None


def linear_fp8(x32: Tensor, w: Tensor, b: Optional[Tensor], k: Optional[int] = None, via_bf16: bool = False,
               gscale: Optional[float] = None) -> Tensor:
    """Block Linear on e4m3 operands.  `x32` is the fp32 value the HIP producer holds when it writes its bf16 and
    e4m3 images (via_bf16: the e4m3 image is made from the bf16 one).  `k`: scale exponent of the weight tensor
    (default: from this tensor's own max; the fused QKV weight shares one).  `gscale`: scale the HIP path quantised
    the gradient of this layer's output with (exported by the engine); None = bf16 input-gradient product."""
    if k is None:
        k = fp8_scale_exp(float(w.detach().abs().max()))
    y = _LinearFP8.apply(x32, w, k, via_bf16, gscale)
    if b is not None:
        y = y + b
    return y


# --------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------
def patchify(x: Tensor, p: int) -> Tensor:
    """nn.Unfold(k=p, s=p) + permute(0,2,1): [B,C,H,W] -> [B, N, C*p*p].

    Feature order (c, kh, kw), patch order row-major (oh, ow)
    (vit_core/ssl/simmim/model.py:27,43; vit_core/patch_embedding.py:113,123-124).
    """
    B, C, H, W = x.shape
    gh, gw = H // p, W // p
    x = x.reshape(B, C, gh, p, gw, p)
    x = x.permute(0, 2, 4, 1, 3, 5)  # B, gh, gw, C, p, p
    return x.reshape(B, gh * gw, C * p * p)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], emu=None) -> Tensor:
    """y = x W^T + b, fp32 accumulate; operands rounded under emu."""
    y = rnd(x, emu) @ rnd(w, emu).t()
    if b is not None:
        y = y + b
    return y


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(D), biased variance, eps 1e-5 (vit_core/encoder_block.py:26-27)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """F.gelu default = exact erf form (vit_core/feed_forward.py:26)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def softmax_lastdim(s: Tensor) -> Tensor:
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    return e / e.sum(dim=-1, keepdim=True)


# emu="bf16" only: the BACKWARD of sdpa() as every flash-style kernel (and the HIP path, csrc/attention.hip) computes it,
# dS = P (dP - delta) with delta = rowsum(dO * O) taken from the bf16-STORED output, instead of autograd's softmax backward
# on the materialised P (delta = rowsum(P * dP): every row of dS sums to zero exactly).  The two agree to ~2^-9 |dO||O|;
# where dP is nearly constant over the keys (few, near-identical tokens) dS cancels and the query / key weight gradients
# move by several per cent.  Off by default (the reference's eager path is the autograd form); the parity sweeps switch it
# on with `flash_delta()` to tell that conditioning apart from a kernel error.
_FLASH_DELTA = False


class flash_delta:
    """Context manager: sdpa(emu="bf16") takes its backward in the flash-style form inside the block."""

    def __enter__(self):
        global _FLASH_DELTA
        self._old, _FLASH_DELTA = _FLASH_DELTA, True
        return self

    def __exit__(self, *exc):
        global _FLASH_DELTA
        _FLASH_DELTA = self._old
        return False


def _bf(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _SdpaFlashBwd(torch.autograd.Function):
    """Forward of sdpa(emu="bf16") (q, k, v arrive bf16-rounded); backward on bf16 operands with fp32 accumulation:
    dV = bf16(P)^T dO, dP = dO V^T, delta = rowsum(dO * bf16(O)), dS = bf16(P (dP - delta)), dQ = dS K / sqrt(d),
    dK = dS^T Q / sqrt(d)."""

    @staticmethod
    def forward(ctx, q, k, v):
        sc = 1.0 / math.sqrt(q.shape[-1])
        p = softmax_lastdim((q @ k.transpose(-2, -1)) * sc)
        o = _bf(p) @ v
        ctx.save_for_backward(q, k, v, p, _bf(o))
        ctx.sc = sc
        ctx.mark_non_differentiable(p)
        return o, p

    @staticmethod
    def backward(ctx, do, _dp):
        q, k, v, p, ob = ctx.saved_tensors
        dob = _bf(do)
        dv = _bf(p).transpose(-2, -1) @ dob
        dp = dob @ v.transpose(-2, -1)
        delta = (dob * ob).sum(dim=-1, keepdim=True)
        ds = _bf(p * (dp - delta))
        return (ds @ k) * ctx.sc, (ds.transpose(-2, -1) @ q) * ctx.sc, dv


def sdpa(q: Tensor, k: Tensor, v: Tensor, emu=None) -> Tuple[Tensor, Tensor]:
    """ScaledDotProductAttention (vit_core/attention.py:20-23): no mask, no dropout."""
    if _FLASH_DELTA and emu == "bf16":
        return _SdpaFlashBwd.apply(q, k, v)
    s = q @ k.transpose(-2, -1)
    s = s / math.sqrt(q.shape[-1])
    p = softmax_lastdim(s)
    o = rnd(p, emu) @ v
    return o, p


def mha(x: Tensor, sd: SD, pre: str, num_heads: int, emu=None, return_attn=False, gscales=None):
    """MultiHeadedAttention.forward, self-attention case (vit_core/attention.py:78-106).
    Three bias-free projections, head split, SDPA, merge, bias-free final_linear."""
    B, N, D = x.shape
    dh = D // num_heads
    if emu == "fp8":
        # x is the un-rounded LayerNorm output; the HIP path multiplies by ONE fused [3D, D] weight image
        kq = fp8_scale_exp(max(float(sd[pre + n].detach().abs().max()) for n in ("w_query.weight", "w_key.weight", "w_value.weight")))
        gq = None if gscales is None else float(gscales[3])      # one scale for the fused dQKV tensor
        q = rnd(linear_fp8(x, sd[pre + "w_query.weight"], None, kq, gscale=gq), emu)
        k = rnd(linear_fp8(x, sd[pre + "w_key.weight"], None, kq, gscale=gq), emu)
        v = rnd(linear_fp8(x, sd[pre + "w_value.weight"], None, kq, gscale=gq), emu)
    else:
        q = rnd(linear(x, sd[pre + "w_query.weight"], None, emu), emu)
        k = rnd(linear(x, sd[pre + "w_key.weight"], None, emu), emu)
        v = rnd(linear(x, sd[pre + "w_value.weight"], None, emu), emu)
    q = q.view(B, N, num_heads, dh).transpose(1, 2)
    k = k.view(B, N, num_heads, dh).transpose(1, 2)
    v = v.view(B, N, num_heads, dh).transpose(1, 2)
    o, p = sdpa(q, k, v, "bf16" if emu == "fp8" else emu)
    if emu == "fp8":
        # the attention kernel writes its bf16 and e4m3 images from the same fp32 values
        out = linear_fp8(o.transpose(1, 2).reshape(B, N, D), sd[pre + "final_linear.weight"], None,
                         gscale=None if gscales is None else float(gscales[2]))
    else:
        o = rnd(o, emu).transpose(1, 2).reshape(B, N, D)
        out = linear(o, sd[pre + "final_linear.weight"], None, emu)
    return (out, p) if return_attn else (out, None)


def feed_forward(x: Tensor, sd: SD, pre: str, emu=None, keep_inner: Optional[Tensor] = None,
                 p_drop: float = 0.0, gscales=None) -> Tensor:
    """FeedForwardBlock.forward (vit_core/feed_forward.py:26-28)."""
    if emu == "fp8":
        u = rnd(linear_fp8(x, sd[pre + "linear_in.weight"], sd[pre + "linear_in.bias"],
                           gscale=None if gscales is None else float(gscales[1])), emu)
    else:
        u = rnd(linear(x, sd[pre + "linear_in.weight"], sd[pre + "linear_in.bias"], emu), emu)
    a = gelu_erf(u)
    if keep_inner is not None:
        a = a * keep_inner / (1.0 - p_drop)
    if emu == "fp8":
        return linear_fp8(a, sd[pre + "linear_out.weight"], sd[pre + "linear_out.bias"],
                          gscale=None if gscales is None else float(gscales[0]))
    a = rnd(a, emu)
    return linear(a, sd[pre + "linear_out.weight"], sd[pre + "linear_out.bias"], emu)


def encoder_block(x: Tensor, sd: SD, pre: str, num_heads: int, emu=None,
                  keep: Optional[Sequence[Tensor]] = None, p_drop: float = 0.0,
                  return_attn=False, fp8_gscales=None):
    """EncoderBlock.forward, Pre-LN (vit_core/encoder_block.py:40-53).

    ``keep`` = optional (keep1, keep_inner, keep2) 0/1 masks for the three
    dropout sites (drop1, FFN inner, drop2); survivors scaled by 1/(1-p).
    ``fp8_gscales`` (emu="fp8" only) = the four scales the HIP path quantised this block's gradient operands with
    (d FFN-out, d FFN-hidden, d attention-out, dQKV): the input-gradient products then run on e4m3 operands too."""
    sc = 1.0 / (1.0 - p_drop) if keep is not None else 1.0
    # (fp8: the Linear layers take the un-rounded LayerNorm output and make both operand images themselves)
    ln_emu = None if emu == "fp8" else emu
    h = rnd(layer_norm(x, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"]), ln_emu)
    a, probs = mha(h, sd, pre + "self_attention.", num_heads, emu, return_attn, gscales=fp8_gscales)
    if keep is not None:
        a = a * keep[0] * sc
    x = x + a
    h = rnd(layer_norm(x, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"]), ln_emu)
    f = feed_forward(h, sd, pre + "feed_forward.", emu,
                     keep_inner=None if keep is None else keep[1], p_drop=p_drop, gscales=fp8_gscales)
    if keep is not None:
        f = f * keep[2] * sc
    x = x + f
    return x, probs


def num_blocks_of(sd: SD, pre: str = "encoder_blocks.") -> int:
    idx = set()
    for k in sd:
        if k.startswith(pre):
            idx.add(int(k[len(pre):].split(".")[0]))
    return len(idx)


# --------------------------------------------------------------------------
# SimMIM
# --------------------------------------------------------------------------
def simple_masking(batch: int, num_patches: int, mask_ratio: float,
                   generator: Optional[torch.Generator] = None) -> Tensor:
    """Bool mask [B,N] of vit_core/ssl/simmim/masking.py:21-33: for each image one
    ``torch.randperm(N)[:int(N*ratio)]`` draw, in batch order, from the (CPU)
    generator; scattered into a bool mask."""
    nm = int(num_patches * mask_ratio)
    mask = torch.zeros(batch, num_patches, dtype=torch.bool)
    for b in range(batch):
        perm = torch.randperm(num_patches, generator=generator)
        mask[b, perm[:nm]] = True
    return mask


def simmim_forward(sd: SD, x: Tensor, mask: Tensor, patch: int, num_heads: int, emu=None,
                   keeps: Optional[List[Sequence[Tensor]]] = None, p_drop: float = 0.0, fp8_gscales=None):
    """SimMIMViT.forward (vit_core/ssl/simmim/model.py:43-62) for a given bool mask.
    Returns (pred [B*nm, Pd], targets [B*nm, Pd]); rows in ascending (b, n) order."""
    patches = patchify(x, patch)
    targets = patches[mask]
    tok = linear(patches, sd["projection.weight"], sd["projection.bias"], emu)
    tok = torch.where(mask.unsqueeze(-1), sd["mask_token"], tok)
    tok = tok + sd["positional_embedding"]
    h = tok
    for i in range(num_blocks_of(sd)):
        h, _ = encoder_block(h, sd, f"encoder_blocks.{i}.", num_heads, emu,
                             keep=None if keeps is None else keeps[i], p_drop=p_drop,
                             fp8_gscales=None if fp8_gscales is None else fp8_gscales[i])
    sel = h[mask]
    pred = linear(sel, sd["simmim_head.weight"], sd["simmim_head.bias"], emu)
    return pred, targets


def simmim_inference(sd: SD, x: Tensor, patch: int, num_heads: int, emu=None,
                     return_patch_features=False) -> Tensor:
    """SimMIMViT.inference_forward (vit_core/ssl/simmim/model.py:81-93)."""
    patches = patchify(x, patch)
    h = linear(patches, sd["projection.weight"], sd["projection.bias"], emu) + sd["positional_embedding"]
    for i in range(num_blocks_of(sd)):
        h, _ = encoder_block(h, sd, f"encoder_blocks.{i}.", num_heads, emu)
    return h if return_patch_features else h.mean(dim=1)


def l1_loss_mean(pred: Tensor, target: Tensor) -> Tensor:
    """nn.L1Loss(reduction='mean') (configs/simmim/training.yaml:2-5)."""
    return (pred - target).abs().sum() / pred.numel()


# --------------------------------------------------------------------------
# supervised ViT
# --------------------------------------------------------------------------
def conv_patch_embed(x: Tensor, w: Tensor, b: Tensor, cls: Tensor, pos: Tensor, patch: int, emu=None):
    """ConvolutionalPatchEmbedding.forward (vit_core/patch_embedding.py:91-96):
    Conv2d(k=s=P) == patchify @ W.view(E,-1)^T + b; prepend CLS; += pos."""
    B = x.shape[0]
    tok = linear(patchify(x, patch), w.reshape(w.shape[0], -1), b, emu)
    tok = torch.cat([cls.expand(B, -1, -1), tok], dim=1)
    return tok + pos


def vit_forward(sd: SD, x: Tensor, patch: int, num_heads: int, emu=None, return_attn=False):
    """ViT.forward (vit_core/vit.py:34-45) + MLPHead (vit_core/mlp_head.py:13-14)."""
    h = conv_patch_embed(x, sd["patch_embedding.conv.weight"], sd["patch_embedding.conv.bias"],
                         sd["patch_embedding.cls_token"], sd["patch_embedding.positional_embedding"],
                         patch, emu)
    probs = None
    for i in range(num_blocks_of(sd)):
        h, probs = encoder_block(h, sd, f"encoder_blocks.{i}.", num_heads, emu, return_attn=return_attn)
    c = h[:, 0]
    c = rnd(layer_norm(c, sd["classification_head.norm.weight"], sd["classification_head.norm.bias"]), emu)
    logits = linear(c, sd["classification_head.linear.weight"], sd["classification_head.linear.bias"], emu)
    return (logits, probs) if return_attn else logits


def cross_entropy_mean(logits: Tensor, labels: Tensor) -> Tensor:
    m = logits.max(dim=-1, keepdim=True).values
    lse = m.squeeze(-1) + torch.log(torch.exp(logits - m).sum(dim=-1))
    picked = logits.gather(1, labels.view(-1, 1)).squeeze(1)
    return (lse - picked).mean()


# --------------------------------------------------------------------------
# DINO
# --------------------------------------------------------------------------
def _cubic_w(t: Tensor, a: float = -0.75):
    """Keys cubic convolution coefficients (PyTorch upsample_bicubic2d, A=-0.75)."""
    def c1(x):  # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1
    def c2(x):  # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    return [c2(t + 1), c1(t), c1(1 - t), c2(2 - t)]


def bicubic_resize(img: Tensor, out_h: int, out_w: int) -> Tensor:
    """F.interpolate(mode='bicubic', align_corners=False) on [1,C,H,W]
    (vit_core/patch_embedding.py:41-45), restated with the Keys kernel A=-0.75
    and border-clamped taps, separable (rows then columns)."""
    def resize_axis(t: Tensor, out_n: int, axis: int) -> Tensor:
        in_n = t.shape[axis]
        scale = in_n / out_n
        dst = torch.arange(out_n, dtype=torch.float32)
        src = (dst + 0.5) * scale - 0.5
        i0 = torch.floor(src)
        frac = src - i0
        ws = _cubic_w(frac)
        out = 0
        for tap, wgt in zip((-1, 0, 1, 2), ws):
            idx = (i0.long() + tap).clamp(0, in_n - 1)
            g = t.index_select(axis, idx)
            shape = [1] * t.dim()
            shape[axis] = out_n
            out = out + g * wgt.view(shape)
        return out
    return resize_axis(resize_axis(img, out_h, 2), out_w, 3)


def dynamic_patch_embed(x: Tensor, sd: SD, pre: str, patch: int, grid: Tuple[int, int], emu=None):
    """DynamicPatchEmbedding.forward + interpolate_pos_encoding
    (vit_core/patch_embedding.py:26-63).  Note the reference passes (w,h) =
    (grid rows, grid cols) of the *conv output* as the interpolate size."""
    B = x.shape[0]
    w_ = sd[pre + "proj.weight"]
    tok = linear(patchify(x, patch), w_.reshape(w_.shape[0], -1), sd[pre + "proj.bias"], emu)
    gh, gw = x.shape[2] // patch, x.shape[3] // patch
    pos = sd[pre + "positional_embedding"]
    D = pos.shape[-1]
    if not (gh * gw == grid[0] * grid[1] and gh == gw):
        cls_pos = pos[:, :1]
        pp = pos[:, 1:].reshape(1, grid[0], grid[1], D).permute(0, 3, 1, 2)
        pp = bicubic_resize(pp, gh, gw)
        pp = pp.permute(0, 2, 3, 1).reshape(1, -1, D)
        pos = torch.cat([cls_pos, pp], dim=1)
    tok = torch.cat([sd[pre + "cls_token"].expand(B, -1, -1), tok], dim=1)
    return tok + pos


def dino_backbone(sd: SD, pre: str, x: Tensor, patch: int, num_heads: int, grid, emu=None):
    """ViTBackbone.forward (vit_core/ssl/dino/model.py:34-45): CLS token output."""
    h = dynamic_patch_embed(x, sd, pre + "patch_embedding.", patch, grid, emu)
    for i in range(num_blocks_of(sd, pre + "encoder_blocks.")):
        h, _ = encoder_block(h, sd, pre + f"encoder_blocks.{i}.", num_heads, emu)
    return h[:, 0]


def dino_head(sd: SD, pre: str, x: Tensor, emu=None) -> Tensor:
    """DINOHead.forward (vit_core/ssl/dino/head.py:19-23): MLP(GELU) -> L2
    normalise rows (eps 1e-12) -> weight-normed Linear, W = g * v / ||v||_row."""
    h = rnd(gelu_erf(rnd(linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"], emu), emu)), emu)
    h = rnd(gelu_erf(rnd(linear(h, sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"], emu), emu)), emu)
    h = linear(h, sd[pre + "mlp.4.weight"], sd[pre + "mlp.4.bias"], emu)
    nrm = torch.sqrt((h * h).sum(dim=1, keepdim=True)).clamp_min(1e-12)
    h = h / nrm
    g = sd[pre + "fully_connected.parametrizations.weight.original0"]
    v = sd[pre + "fully_connected.parametrizations.weight.original1"]
    w = v * (g / torch.sqrt((v * v).sum(dim=1, keepdim=True)))
    return linear(h, w, sd[pre + "fully_connected.bias"], emu)


def dino_forward(sd: SD, views: List[Tensor], num_global: int, patch: int, num_heads: int,
                 grid, center: Tensor, center_momentum: float, emu=None):
    """DINOViT.forward (vit_core/ssl/dino/model.py:110-124) + _update_center (:91-99).
    Returns (teacher_out [G*B,K], student_out [V*B,K], new_center [1,K])."""
    g = torch.cat(views[:num_global], dim=0)
    l = torch.cat(views[num_global:], dim=0)
    s_g = dino_head(sd, "student_head.", dino_backbone(sd, "student_backbone.", g, patch, num_heads, grid, emu), emu)
    s_l = dino_head(sd, "student_head.", dino_backbone(sd, "student_backbone.", l, patch, num_heads, grid, emu), emu)
    student = torch.cat([s_g, s_l])
    with torch.no_grad():
        teacher = dino_head(sd, "teacher_head.", dino_backbone(sd, "teacher_backbone.", g, patch, num_heads, grid, emu), emu)
        new_center = center_momentum * center + (1 - center_momentum) * teacher.mean(dim=0)
    return teacher, student, new_center


def dino_loss_naive(teacher: Tensor, student: Tensor, center: Tensor, t_temp: float, s_temp: float) -> Tensor:
    """DINOLoss.forward exactly as written (vit_core/ssl/dino/loss.py:22-29):
    teacher [G,B,K], student [V,B,K] -> -(t[:,None]*s[None]).sum(1).mean()."""
    s = student / s_temp
    s = s - s.max(dim=-1, keepdim=True).values
    s = s - torch.log(torch.exp(s).sum(dim=-1, keepdim=True))
    t = softmax_lastdim((teacher.detach() - center) / t_temp)
    return -(t.unsqueeze(1) * s.unsqueeze(0)).sum(dim=1).mean()


def dino_loss_algebraic(teacher: Tensor, student: Tensor, center: Tensor, t_temp: float, s_temp: float) -> Tensor:
    """Same value via loss = -(1/(G*B*K)) sum_{b,k} (sum_g t)(sum_v s)."""
    G, B, K = teacher.shape
    s = student / s_temp
    s = s - s.max(dim=-1, keepdim=True).values
    s = s - torch.log(torch.exp(s).sum(dim=-1, keepdim=True))
    t = softmax_lastdim((teacher.detach() - center) / t_temp)
    return -(t.sum(dim=0) * s.sum(dim=0)).sum() / (G * B * K)


def ema_update(teacher: Tensor, student: Tensor, m: float) -> Tensor:
    """momentum_update_teacher per tensor (vit_core/ssl/dino/model.py:131-133)."""
    return teacher * m + (1 - m) * student


def dino_momentum(step: int, m_start: float, m_end: float, total: int) -> float:
    """DINOMomentumScheduler.get_momentum (vit_core/ssl/dino/dino_utils.py:10-14)."""
    if step >= total:
        return m_end
    return m_end - (m_end - m_start) * 0.5 * (1 + math.cos(math.pi * step / total))


def dino_teacher_temp(step: int, t_start: float, t_end: float, total: int, kind: str = "cosine") -> float:
    """DINOTeacherTempScheduler.get_temp (vit_core/ssl/dino/dino_utils.py:29-36)."""
    if step >= total:
        return t_end
    prog = step / total
    if kind == "linear":
        return t_start + (t_end - t_start) * prog
    return t_end - (t_end - t_start) * 0.5 * (1 + math.cos(math.pi * prog))


# --------------------------------------------------------------------------
# optimizer step (the reference builds torch.optim.AdamW reflectively,
# utils/train_utils.py:25-29; configs/base/training.yaml:10-15)
# --------------------------------------------------------------------------
def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1=0.9, beta2=0.999, eps=1e-8, wd=1e-2):
    """One decoupled-weight-decay Adam step (torch.optim.AdamW semantics,
    amsgrad=False, maximize=False).  ``step`` is 1-based.  Returns (p, m, v)."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def linear_warmup_lr(step: int, warmup_steps: int, start_lr: float, target_lr: float) -> float:
    """LinearWarmupScheduler.step (utils/schedulers.py:12-19), lr after `step` calls."""
    warmup_steps = max(1, warmup_steps)
    return start_lr + (float(step) / warmup_steps) * (target_lr - start_lr)


# --------------------------------------------------------------------------
# full training step used by bench.py's cpu_baseline leg
# --------------------------------------------------------------------------
def simmim_train_step(sd: SD, opt_state: Dict[str, Tuple[Tensor, Tensor]], x: Tensor, mask: Tensor,
                      patch: int, num_heads: int, step: int, lr: float, wd: float,
                      p_drop: float = 0.0, generator: Optional[torch.Generator] = None) -> float:
    """zero_grad -> forward -> L1 -> backward -> AdamW, eager fp32
    (utils/trainers/simmim_trainer.py:61-77 on a CPU host)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    keeps = None
    if p_drop > 0:
        B = x.shape[0]
        N = mask.shape[1]
        D = sd["positional_embedding"].shape[-1]
        F = sd["encoder_blocks.0.feed_forward.linear_in.weight"].shape[0]
        keeps = []
        for _ in range(num_blocks_of(sd)):
            keeps.append(tuple((torch.rand(B, N, d, generator=generator) >= p_drop).float() for d in (D, F, D)))
    pred, tgt = simmim_forward(leaves, x, mask, patch, num_heads, keeps=keeps, p_drop=p_drop)
    loss = l1_loss_mean(pred, tgt)
    loss.backward()
    with torch.no_grad():
        for k in sd:
            m, v = opt_state.setdefault(k, (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])))
            p2, m2, v2 = adamw_step(sd[k], leaves[k].grad, m, v, step, lr, wd=wd)
            sd[k] = p2
            opt_state[k] = (m2, v2)
    return float(loss)
