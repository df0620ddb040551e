"""torch.autograd glue for STAND-ALONE use of the drop-in modules (EncoderBlock,
MultiHeadedAttention, FeedForwardBlock, ScaledDotProductAttention, MLPHead, the patch
embeddings).  The full models (ViT / SimMIMViT / DINOViT) do not go through these: they
run one explicit forward/backward schedule in vitssl_hip.engine.

Every Function stages bf16 GEMM operands, calls the HIP kernels through the C ABI and
returns fp32 tensors, so the modules compose with ordinary PyTorch code."""
import torch
from torch.autograd import Function

from . import _runtime as R
from ._runtime import BF16, F32, L, ops


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _round_up(n, a):
    return (n + a - 1) // a * a


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, cols] -> bf16 through the HIP cast kernel."""
    if x.dtype == BF16:
        return x.contiguous()
    x = R.as_f32(x)
    out = _empty(x.shape, BF16, x)
    ops.cast_bf16(x, out)
    return out


def weight_bf16(w: torch.Tensor, want_t=True):
    """fp32 [N,K] -> (bf16 [Np,K], bf16 [K,Np]) with N padded to a multiple of 8 (zeros)."""
    N, K = w.shape
    if K % 64 != 0:
        raise L.VitsslError(f"linear layer with in_features={K}: the MFMA GEMM needs a multiple of 64")
    Np = _round_up(N, 8)
    src = R.as_f32(w.detach())
    if Np != N:
        pad = torch.zeros(Np, K, dtype=F32, device=w.device)
        pad[:N] = src
        src = pad
    wb = _empty((Np, K), BF16, w)
    wt = _empty((K, Np), BF16, w) if want_t else None
    ops.cast_transpose_bf16(src, wb, wt)
    return wb, wt


def _pad_cols_bf16(x: torch.Tensor, Np: int) -> torch.Tensor:
    if x.shape[1] == Np:
        return x
    out = torch.zeros(x.shape[0], Np, dtype=x.dtype, device=x.device)
    out[:, :x.shape[1]] = x
    return out


def _colsum(dy2: torch.Tensor) -> torch.Tensor:
    """Column sums of an fp32 [rows, cols] gradient (bias gradient)."""
    rows, cols = dy2.shape
    if cols % 4 == 0 and cols <= 2048:
        scratch = _empty((rows, cols), BF16, dy2)
        cs = torch.zeros(cols, dtype=F32, device=dy2.device)
        ops.grad_mask_cast(dy2, scratch, cs)
        return cs
    return dy2.sum(0)


# --------------------------------------------------------------------------- linear
class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, b):
        lead = x.shape[:-1]
        K = x.shape[-1]
        N = w.shape[0]
        xb = to_bf16(x.reshape(-1, K))
        wb, wt = weight_bf16(w)
        Np = wb.shape[0]
        bias = None
        if b is not None:
            bias = torch.zeros(Np, dtype=F32, device=x.device)
            bias[:N] = b.detach().float()
        y = _empty((xb.shape[0], Np), F32, x)
        ops.gemm_nt(xb, wb, y, L.EPI_F32, bias=bias)
        ctx.save_for_backward(xb, wt)
        ctx.meta = (lead, K, N, Np, b is not None)
        return y[:, :N].reshape(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        xb, wt = ctx.saved_tensors
        lead, K, N, Np, has_b = ctx.meta
        dy2 = R.as_f32(dy.reshape(-1, N))
        dyb = _pad_cols_bf16(to_bf16(dy2), Np)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if Np % 64 == 0:
                dx = _empty((dy2.shape[0], K), F32, dy2)
                ops.gemm_nt(dyb, wt, dx, L.EPI_F32)
            else:  # tiny output layer (e.g. 10 classes): pad the contraction to 64
                Nk = _round_up(Np, 64)
                wt2 = torch.zeros(K, Nk, dtype=BF16, device=dy2.device)
                wt2[:, :Np] = wt
                dx = _empty((dy2.shape[0], K), F32, dy2)
                ops.gemm_nt(_pad_cols_bf16(dyb, Nk), wt2, dx, L.EPI_F32)
            dx = dx.reshape(*lead, K)
        if ctx.needs_input_grad[1]:
            dwp = torch.zeros(Np, K, dtype=F32, device=dy2.device)
            ops.gemm_tn(dyb, xb, dwp)
            dw = dwp[:N]
        if has_b and ctx.needs_input_grad[2]:
            db = _colsum(dy2)
        return dx, dw, db


def linear_apply(x, w, b=None):
    return _LinearFn.apply(x, w, b)


# --------------------------------------------------------------------------- LayerNorm + Linear
class _LnLinearFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, w, b, eps):
        lead = x.shape[:-1]
        D = x.shape[-1]
        N = w.shape[0]
        x2 = R.as_f32(x.reshape(-1, D))
        rows = x2.shape[0]
        h = _empty((rows, D), BF16, x)
        mean = _empty((rows,), F32, x)
        rstd = _empty((rows,), F32, x)
        ops.layernorm_fwd(x2, R.as_f32(gamma.detach()), R.as_f32(beta.detach()), h, mean, rstd, eps)
        wb, wt = weight_bf16(w)
        Np = wb.shape[0]
        bias = torch.zeros(Np, dtype=F32, device=x.device)
        if b is not None:
            bias[:N] = b.detach().float()
        y = _empty((rows, Np), F32, x)
        ops.gemm_nt(h, wb, y, L.EPI_F32, bias=bias)
        ctx.save_for_backward(x2, h, mean, rstd, gamma, wb)
        ctx.meta = (lead, D, N, Np, b is not None)
        return y[:, :N].reshape(*lead, N)

    @staticmethod
    def backward(ctx, dy):
        x2, h, mean, rstd, gamma, wb = ctx.saved_tensors
        lead, D, N, Np, has_b = ctx.meta
        rows = x2.shape[0]
        dy2 = R.as_f32(dy.reshape(-1, N))
        Nk = _round_up(Np, 64)
        dyb = _pad_cols_bf16(to_bf16(dy2), Nk)
        # dh = dy . W  (B operand = W^T [D, Nk])
        wt = torch.zeros(D, Nk, dtype=BF16, device=dy2.device)
        wt[:, :Np] = wb.t()
        dh = _empty((rows, D), BF16, dy2)
        ops.gemm_nt(dyb, wt, dh, L.EPI_BF16)
        dx = _empty((rows, D), F32, dy2)
        dgamma = torch.zeros(D, dtype=F32, device=dy2.device)
        dbeta = torch.zeros(D, dtype=F32, device=dy2.device)
        ops.layernorm_bwd(dh, x2, mean, rstd, R.as_f32(gamma.detach()), None, dx, None, dgamma, dbeta)
        dwp = torch.zeros(Nk, D, dtype=F32, device=dy2.device)
        ops.gemm_tn(dyb, h, dwp)
        db = _colsum(dy2) if has_b else None
        return dx.reshape(*lead, D), dgamma, dbeta, dwp[:N], db, None


def ln_linear_apply(x, gamma, beta, w, b, eps=1e-5):
    return _LnLinearFn.apply(x, gamma, beta, w, b, eps)


# --------------------------------------------------------------------------- FFN
class _FFNFn(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p):
        lead = x.shape[:-1]
        D = x.shape[-1]
        Fd = w1.shape[0]
        if D % 8 or Fd % 8:
            raise L.VitsslError("FeedForwardBlock: d_model and d_ff must be multiples of 8")
        xb = to_bf16(x.reshape(-1, D))
        rows = xb.shape[0]
        w1b, w1t = weight_bf16(w1)
        w2b, w2t = weight_bf16(w2)
        drop = ops.make_dropout(p, R.next_seed(), 1) if p > 0 else ops.NO_DROP
        u = _empty((rows, Fd), BF16, x)
        a = _empty((rows, Fd), BF16, x)
        ops.gemm_nt(xb, w1b, u, L.EPI_GELU, bias=R.as_f32(b1.detach()), out1=a, drop=drop)
        y = _empty((rows, D), F32, x)
        ops.gemm_nt(a, w2b, y, L.EPI_F32, bias=R.as_f32(b2.detach()))
        ctx.save_for_backward(xb, u, a, w1t, w2t)
        ctx.meta = (lead, D, Fd, drop)
        return y.reshape(*lead, D)

    @staticmethod
    def backward(ctx, dy):
        xb, u, a, w1t, w2t = ctx.saved_tensors
        lead, D, Fd, drop = ctx.meta
        dy2 = R.as_f32(dy.reshape(-1, D))
        rows = dy2.shape[0]
        dyb = _empty((rows, D), BF16, dy2)
        db2 = torch.zeros(D, dtype=F32, device=dy2.device)
        ops.grad_mask_cast(dy2, dyb, db2)
        du = _empty((rows, Fd), BF16, dy2)
        db1 = torch.zeros(Fd, dtype=F32, device=dy2.device)
        ops.gemm_nt(dyb, w2t, du, L.EPI_DGELU, aux=u, colsum=db1)   # u holds g' (mask and scale folded in)
        dw2 = torch.zeros(D, Fd, dtype=F32, device=dy2.device)
        ops.gemm_tn(dyb, a, dw2)
        dx = _empty((rows, D), F32, dy2)
        ops.gemm_nt(du, w1t, dx, L.EPI_F32)
        dw1 = torch.zeros(Fd, D, dtype=F32, device=dy2.device)
        ops.gemm_tn(du, xb, dw1)
        return dx.reshape(*lead, D), dw1, db1, dw2, db2, None


def ffn_apply(x, w1, b1, w2, b2, p):
    return _FFNFn.apply(x, w1, b1, w2, b2, float(p))


# --------------------------------------------------------------------------- attention
def _check_attn_geometry(q, k, v):
    if not (q.shape == k.shape == v.shape):
        raise L.VitsslError(f"attention: query/key/value shapes {tuple(q.shape)}/{tuple(k.shape)}/{tuple(v.shape)} "
                            "differ; the fused kernel covers the self-attention geometry only")


class _SDPAFn(Function):
    """q, k, v: [B, H, N, dh] (fp32 in/out)."""

    @staticmethod
    def forward(ctx, q, k, v, return_attn):
        Bn, H, N, dh = q.shape
        qkv = torch.stack([q, k, v], dim=2)                      # [B,H,3,N,dh]
        qkv = to_bf16(R.as_f32(qkv.permute(0, 3, 2, 1, 4)).reshape(Bn * N, 3 * H * dh))
        out = _empty((Bn * N, H * dh), BF16, q)
        lse = _empty((Bn, H, N), F32, q)
        probs = _empty((Bn, H, N, N), F32, q) if return_attn else None
        ops.attn_fwd(qkv, out, lse, Bn, N, H, dh, probs=probs)
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (Bn, H, N, dh)
        o = out.float().view(Bn, N, H, dh).transpose(1, 2)
        if return_attn:
            ctx.mark_non_differentiable(probs)
            return o, probs
        return o, None

    @staticmethod
    def backward(ctx, do, _dp):
        qkv, out, lse = ctx.saved_tensors
        Bn, H, N, dh = ctx.meta
        dout = to_bf16(R.as_f32(do.transpose(1, 2)).reshape(Bn * N, H * dh))
        dqkv = _empty(qkv.shape, BF16, qkv)
        delta = _empty((Bn, H, N), F32, qkv)
        ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, Bn, N, H, dh)
        d = dqkv.float().view(Bn, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        return d[0], d[1], d[2], None


def sdpa_apply(q, k, v, return_attn=False):
    _check_attn_geometry(q, k, v)
    shape = q.shape
    if q.dim() < 2:
        raise L.VitsslError("attention: expected [..., seq, d_k] tensors")
    N, dh = shape[-2], shape[-1]
    lead = shape[:-2]
    if q.dim() == 4:
        q4, k4, v4 = q, k, v
    else:  # collapse leading dims into a batch of single-head problems
        q4, k4, v4 = (t.reshape(-1, 1, N, dh) for t in (q, k, v))
    o, p = _SDPAFn.apply(q4.float(), k4.float(), v4.float(), return_attn)
    o = o.reshape(*lead, N, dh).to(q.dtype)
    if return_attn:
        p = p.reshape(*lead, N, N)
    return o, p


class _MHAFn(Function):
    @staticmethod
    def forward(ctx, query, key, value, wq, wk, wv, wo, H, return_attn):
        Bn, N, D = query.shape
        dh = D // H
        same = (key is query or key.data_ptr() == query.data_ptr()) and (value is query or value.data_ptr() == query.data_ptr())
        rows = Bn * N
        if same:
            xb = to_bf16(query.reshape(rows, D))
            wcat = torch.cat([wq.detach(), wk.detach(), wv.detach()], dim=0)
            wb, wt = weight_bf16(wcat)
            qkv = _empty((rows, 3 * D), BF16, query)
            ops.gemm_nt(xb, wb, qkv, L.EPI_BF16)
            ins = (xb, xb, xb)
            wts = (wt,)
        else:
            ins, parts, wts = [], [], []
            for x, w in ((query, wq), (key, wk), (value, wv)):
                xb = to_bf16(x.reshape(rows, D))
                wb, wt = weight_bf16(w)
                part = _empty((rows, D), BF16, query)
                ops.gemm_nt(xb, wb, part, L.EPI_BF16)
                ins.append(xb)
                parts.append(part)
                wts.append(wt)
            qkv = torch.cat(parts, dim=1).contiguous()
        att = _empty((rows, D), BF16, query)
        lse = _empty((Bn, H, N), F32, query)
        probs = _empty((Bn, H, N, N), F32, query) if return_attn else None
        ops.attn_fwd(qkv, att, lse, Bn, N, H, dh, probs=probs)
        wob, wot = weight_bf16(wo)
        y = _empty((rows, D), F32, query)
        ops.gemm_nt(att, wob, y, L.EPI_F32)
        ctx.save_for_backward(qkv, att, lse, wot, *ins, *wts)
        ctx.meta = (Bn, N, D, H, same)
        y = y.view(Bn, N, D)
        if return_attn:
            ctx.mark_non_differentiable(probs)
            return y, probs
        return y, None

    @staticmethod
    def backward(ctx, dy, _dp):
        Bn, N, D, H, same = ctx.meta
        sv = ctx.saved_tensors
        qkv, att, lse, wot = sv[:4]
        ins = sv[4:7]
        wts = sv[7:]
        rows = Bn * N
        dh = D // H
        dyb = to_bf16(R.as_f32(dy).reshape(rows, D))
        datt = _empty((rows, D), BF16, dyb)
        ops.gemm_nt(dyb, wot, datt, L.EPI_BF16)
        dwo = torch.zeros(D, D, dtype=F32, device=dyb.device)
        ops.gemm_tn(dyb, att, dwo)
        dqkv = _empty((rows, 3 * D), BF16, dyb)
        delta = _empty((Bn, H, N), F32, dyb)
        ops.attn_bwd(qkv, att, datt, lse, dqkv, delta, Bn, N, H, dh)
        if same:
            dx = _empty((rows, D), F32, dyb)
            ops.gemm_nt(dqkv, wts[0], dx, L.EPI_F32)
            dw = torch.zeros(3 * D, D, dtype=F32, device=dyb.device)
            ops.gemm_tn(dqkv, ins[0], dw)
            dx = dx.view(Bn, N, D)
            # query/key/value are the same tensor: autograd sums the three slots
            return dx, torch.zeros_like(dx), torch.zeros_like(dx), dw[:D], dw[D:2 * D], dw[2 * D:], dwo, None, None
        dxs, dws = [], []
        for i in range(3):
            part = dqkv[:, i * D:(i + 1) * D].contiguous()
            dx = _empty((rows, D), F32, dyb)
            ops.gemm_nt(part, wts[i], dx, L.EPI_F32)
            dw = torch.zeros(D, D, dtype=F32, device=dyb.device)
            ops.gemm_tn(part, ins[i], dw)
            dxs.append(dx.view(Bn, N, D))
            dws.append(dw)
        return dxs[0], dxs[1], dxs[2], dws[0], dws[1], dws[2], dwo, None, None


def mha_apply(query, key, value, wq, wk, wv, wo, H, return_attn=False):
    _check_attn_geometry(query, key, value)
    if query.dim() != 3:
        raise L.VitsslError("MultiHeadedAttention: expected [batch, seq, d_model] inputs")
    y, p = _MHAFn.apply(query, key, value, wq, wk, wv, wo, H, return_attn)
    return y.to(query.dtype), p


# --------------------------------------------------------------------------- encoder stack
class _StackFn(Function):
    @staticmethod
    def forward(ctx, runner, x, training, return_attn, need, *params):
        Bn, T, D = x.shape
        st = runner.store
        st.refresh_weights()
        seed = R.next_seed() if (training and runner.stack.p > 0) else 0
        slot = runner.next_slot() if need else "nograd"
        y, probs = runner.stack.forward(R.as_f32(x).reshape(Bn * T, D), Bn, T, training, seed, save=need, slot=slot,
                                        return_attn=return_attn)
        ctx.runner, ctx.slot, ctx.shape = runner, slot, (Bn, T, D)
        ctx.gen = runner.stamp(slot) if need else 0
        y = y.view(Bn, T, D).clone()      # the stack's buffers are reused by the next call
        if return_attn:
            ctx.mark_non_differentiable(probs)
            return y, probs
        return y, None

    @staticmethod
    def backward(ctx, dy, _dp):
        runner = ctx.runner
        Bn, T, D = ctx.shape
        st = runner.store
        R.check_saved_generation("encoder stack", ctx.gen, runner.slot_gen.get(ctx.slot, -1))
        st.gflat.zero_()
        g = R.as_f32(dy).reshape(Bn * T, D).clone()
        g = runner.stack.backward(g, slot=ctx.slot, scale_key="run")    # slot names only rotate buffers here
        runner.release_slot(ctx.slot)
        grads = [st.gview(n, p.shape).clone() if p.requires_grad else None for n, p in zip(st.names, st.params)]
        return (None, g.view(Bn, T, D), None, None, None, *grads)


class StackRunner:
    """A private FlatStore + EncoderStack for a module that owns encoder blocks."""

    def __init__(self, module, block_prefixes, D, H, Fd, p, device):
        self.module = module
        self.store = R.FlatStore(module, device)
        self.stack = R.EncoderStack(self.store, block_prefixes, D, H, Fd, p)
        self.device = device
        self._slots = 0
        self._gen = 0
        self.slot_gen = {}     # slot -> id of the forward whose activations it holds

    def stamp(self, slot):
        self._gen += 1
        self.slot_gen[slot] = self._gen
        return self._gen

    def valid_for(self, device) -> bool:
        return device == self.device and self.store.is_attached()

    def next_slot(self):
        self._slots += 1
        return f"s{self._slots % 4}"   # a few concurrent graphs (e.g. two views) may be alive

    def release_slot(self, slot):
        pass

    def __call__(self, x, training, return_attn=False):
        need = torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in self.store.params))
        y, p = _StackFn.apply(self, x, training, return_attn, need, *self.store.params)
        return y.to(x.dtype), p


# --------------------------------------------------------------------------- patch embedding
class _PatchEmbedFn(Function):
    """img [B,C,H,W], w [D, C*P*P], b [D], cls [1,1,D], pos [T_out, D] -> tokens [B, T_out, D]."""

    @staticmethod
    def forward(ctx, img, w, b, cls, pos, P):
        img = R.as_f32(img)
        Bn, Cc, Hh, Ww = img.shape
        D, Pd = w.shape
        tokens = (Hh // P) * (Ww // P)
        T = tokens + 1
        if pos.shape[0] != T:
            raise L.VitsslError(f"positional embedding has {pos.shape[0]} rows, input needs {T}")
        patches = _empty((Bn * tokens, Pd), BF16, img)
        ops.patchify_bf16(img, patches, P)
        wb, _ = weight_bf16(w, want_t=False)
        if wb.shape[0] != D:
            raise L.VitsslError("patch embedding: embed dim must be a multiple of 8")
        out = _empty((Bn * T, D), F32, img)
        posf = R.as_f32(pos.detach())
        ops.gemm_nt(patches, wb, out, L.EPI_EMBED, bias=R.as_f32(b.detach()), embed=(None, None, posf, tokens, T, 1))
        out = out.view(Bn, T, D)
        out[:, 0] = cls.detach().reshape(D).float() + posf[0]
        ctx.save_for_backward(patches)
        ctx.meta = (Bn, tokens, T, D, Pd)
        return out

    @staticmethod
    def backward(ctx, dout):
        (patches,) = ctx.saved_tensors
        Bn, tokens, T, D, Pd = ctx.meta
        d = R.as_f32(dout).reshape(Bn * T, D)
        dev = d.device
        dproj = _empty((Bn * tokens, D), BF16, d)
        dpos = torch.zeros(T, D, dtype=F32, device=dev)
        dbias = torch.zeros(D, dtype=F32, device=dev)
        dcls = torch.zeros(D, dtype=F32, device=dev)
        ops.embed_bwd(d, None, dproj, dpos, None, dbias, dcls, Bn, tokens, 1, D)
        dw = torch.zeros(D, Pd, dtype=F32, device=dev)
        ops.gemm_tn(dproj, patches, dw)
        return None, dw, dbias, dcls.view(1, 1, D), dpos, None


def patch_embed_apply(img, w, b, cls, pos, P):
    return _PatchEmbedFn.apply(img, w, b, cls, pos, P)


class _BicubicRowsFn(Function):
    """Bicubic resize of a channel-last token grid [gh0*gw0, D] -> [gh*gw, D] (the patch rows
    of a positional table), ATen-compatible, forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, rows, gh0, gw0, gh, gw):
        rows = R.as_f32(rows)
        out = _empty((gh * gw, rows.shape[1]), F32, rows)
        ops.bicubic_resize_fwd(rows, out, gh0, gw0, gh, gw)
        ctx.dims = (gh0, gw0, gh, gw)
        return out

    @staticmethod
    def backward(ctx, gout):
        gh0, gw0, gh, gw = ctx.dims
        gout = R.as_f32(gout)
        gin = torch.zeros(gh0 * gw0, gout.shape[1], dtype=F32, device=gout.device)
        ops.bicubic_resize_bwd(gout, gin, gh0, gw0, gh, gw)
        return gin, None, None, None, None


def bicubic_rows_apply(rows, src_grid, dst_grid):
    return _BicubicRowsFn.apply(rows, src_grid[0], src_grid[1], dst_grid[0], dst_grid[1])
