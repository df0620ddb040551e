"""Supervised ViT (reference: vit_core/vit.py:9-45): ConvolutionalPatchEmbedding ->
encoder blocks -> CLS token -> MLPHead, run as one engine schedule on a flat store."""
import torch
from torch import nn
from torch.autograd import Function

from . import _runtime as R
from ._runtime import BF16, F32, L, ops
from ._backbone import BackboneRuntime
from .encoder_block import EncoderBlock
from .mlp_head import MLPHead
from .patch_embedding import ConvolutionalPatchEmbedding


def _round_up(n, a):
    return (n + a - 1) // a * a


class _ViTRuntime:
    def __init__(self, model: "ViT", device):
        self.model, self.device = model, device
        C, Hh, Ww = model.input_shape
        P = model.patch_size
        self.store = R.FlatStore(model, device)
        st = self.store
        self.D = model.embed_dim
        self.bb = BackboneRuntime(
            st, "", dict(weight="patch_embedding.conv.weight", bias="patch_embedding.conv.bias",
                         cls="patch_embedding.cls_token", pos="patch_embedding.positional_embedding"),
            len(model.encoder_blocks), C, P, (Hh // P, Ww // P), self.D, model.num_heads, model.mlp_dim, model.dropout_p)
        self.ncls = model.num_classes
        self.rec = None
        self.save_gen = 0

    def valid_for(self, device):
        return device == self.device and self.store.is_attached()

    def _head_weights(self):
        """bf16 copies of the (tiny) classifier weight, zero-padded to the GEMM granules."""
        st, D, C = self.store, self.D, self.ncls
        Nk = _round_up(C, 64)
        w = torch.zeros(Nk, D, dtype=F32, device=self.device)
        w[:C] = st.view("classification_head.linear.weight", (C, D))
        wb = torch.empty(Nk, D, dtype=BF16, device=self.device)
        wt = torch.empty(D, Nk, dtype=BF16, device=self.device)
        ops.cast_transpose_bf16(w, wb, wt)
        return wb, wt, Nk

    def forward(self, x, training, save, return_attn=False):
        st = self.store
        st.refresh_weights()
        seed = R.next_seed() if (training and self.bb.stack.p > 0) else 0
        feats, probs = self.bb.forward(x, training, seed, save=save, slot="a", return_attn=return_attn)
        B = feats.shape[0]
        dev = feats.device
        h = torch.empty(B, self.D, dtype=BF16, device=dev)
        mean = torch.empty(B, dtype=F32, device=dev)
        rstd = torch.empty(B, dtype=F32, device=dev)
        ops.layernorm_fwd(feats, st.view("classification_head.norm.weight"), st.view("classification_head.norm.bias"), h, mean, rstd)
        wb, wt, Nk = self._head_weights()
        bias = torch.zeros(Nk, dtype=F32, device=dev)
        bias[:self.ncls] = st.view("classification_head.linear.bias")
        logits = torch.empty(B, Nk, dtype=F32, device=dev)
        ops.gemm_nt(h, wb, logits, L.EPI_F32, bias=bias)
        if save:
            self.save_gen += 1
            self.rec = dict(feats=feats, h=h, mean=mean, rstd=rstd, wt=wt, Nk=Nk, B=B)
        return logits[:, :self.ncls].contiguous(), probs

    def backward(self, dlogits, reducer=None):
        st, rec = self.store, self.rec
        B, Nk, C, D = rec["B"], rec["Nk"], self.ncls, self.D
        dev = dlogits.device
        gv = st.gview
        dl = torch.zeros(B, Nk, dtype=F32, device=dev)
        dl[:, :C] = R.as_f32(dlogits)
        dlb = torch.empty(B, Nk, dtype=BF16, device=dev)
        ops.cast_bf16(dl, dlb)
        gv("classification_head.linear.bias").add_(dl[:, :C].sum(0))
        dw = torch.zeros(Nk, D, dtype=F32, device=dev)
        ops.gemm_tn(dlb, rec["h"], dw)
        gv("classification_head.linear.weight", (C, D)).add_(dw[:C])
        dh = torch.empty(B, D, dtype=BF16, device=dev)
        ops.gemm_nt(dlb, rec["wt"], dh, L.EPI_BF16)
        dfeats = torch.empty(B, D, dtype=F32, device=dev)
        ops.layernorm_bwd(dh, rec["feats"], rec["mean"], rec["rstd"], st.view("classification_head.norm.weight"), None, dfeats,
                          None, gv("classification_head.norm.weight"), gv("classification_head.norm.bias"))
        if reducer is not None:
            reducer.ready(*st.span("classification_head.norm.weight", "classification_head.linear.bias"))
        self.bb.backward(dfeats, "a", reducer)
        if reducer is not None:
            reducer.ready(*st.span("patch_embedding.cls_token", "patch_embedding.conv.bias"))


class _ViTFn(Function):
    @staticmethod
    def forward(ctx, rt, x, training, return_attn, need, *params):
        logits, probs = rt.forward(x, training, save=need, return_attn=return_attn)
        ctx.rt, ctx.gen = rt, rt.save_gen
        if return_attn:
            ctx.mark_non_differentiable(probs)
            return logits, probs
        return logits, None

    @staticmethod
    def backward(ctx, dlogits, _dp):
        rt = ctx.rt
        R.check_saved_generation("ViT", ctx.gen, rt.save_gen)
        st = rt.store
        st.gflat.zero_()
        rt.backward(dlogits)
        grads = [st.gview(n, p.shape).clone() if p.requires_grad else None for n, p in zip(st.names, st.params)]
        return (None, None, None, None, None, *grads)


class ViT(nn.Module):
    def __init__(
        self,
        num_classes: int,
        num_blocks: int,
        input_shape,
        embed_dim: int,
        patch_size: int,
        num_heads: int = 8,
        mlp_dim: int = 3072,
        dropout: float = 0.1,
    ):
        super().__init__()
        self.encoder_blocks = nn.ModuleList(
            [EncoderBlock(embed_dim, num_heads, mlp_dim, dropout) for _ in range(num_blocks)]
        )
        self.patch_embedding = ConvolutionalPatchEmbedding(input_shape, embed_dim, patch_size)
        self.classification_head = MLPHead(embed_dim, num_classes)
        self.input_shape = tuple(input_shape)
        self.num_classes, self.embed_dim, self.patch_size = num_classes, embed_dim, patch_size
        self.num_heads, self.mlp_dim, self.dropout_p = num_heads, mlp_dim, float(dropout)
        self._rt = None

    def runtime(self, device=None) -> _ViTRuntime:
        device = device or self.patch_embedding.cls_token.device
        if device.type != "cuda":
            raise L.VitsslError("ViT: parameters are on the CPU; move the model to 'cuda' (no CPU fallback)")
        if self._rt is None or not self._rt.valid_for(device):
            L.lib()
            object.__setattr__(self, "_rt", _ViTRuntime(self, device))
        return self._rt

    def flat_store(self):
        return self.runtime().store

    def forward(self, x: torch.Tensor, return_attn=False):
        R.require_gpu(x, "ViT")
        rt = self.runtime(x.device)
        need = torch.is_grad_enabled() and any(p.requires_grad for p in rt.store.params)
        logits, probs = _ViTFn.apply(rt, x, self.training, return_attn, need, *rt.store.params)
        if return_attn:
            return logits, probs
        return logits
