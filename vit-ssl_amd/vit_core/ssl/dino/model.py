"""DINOViT on the MI355X HIP engine (reference: vit_core/ssl/dino/model.py:12-155).

Student and teacher each live in their own flat parameter store with identical layouts,
so the momentum update of ~158 tensors is ONE kernel over the flat buffers, and the
student's data-parallel gradient all-reduce runs over one flat gradient buffer.  The centre
is updated inside forward (before the loss sees it, also in eval), as in the reference;
under data parallelism its batch mean is all-reduced so every rank keeps the same centre.
"""
import copy
from typing import List

import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function

from ... import _runtime as R
from ..._runtime import BF16, F32, L, ops
from ..._backbone import BackboneRuntime
from ...encoder_block import EncoderBlock
from ...patch_embedding import DynamicPatchEmbedding
from ._head_runtime import HeadRuntime
from .head import DINOHead


class ViTBackbone(nn.Module):
    """Encoder blocks + DynamicPatchEmbedding; returns the CLS token
    (reference: ssl/dino/model.py:12-45).  Parameter container inside DINOViT; usable on its
    own through the modules' stand-alone paths."""

    def __init__(self, num_blocks, input_shape, embed_dim, patch_size, num_heads=8, mlp_dim=3072, dropout=0.1):
        super().__init__()
        self.encoder_blocks = nn.ModuleList([EncoderBlock(embed_dim, num_heads, mlp_dim, dropout) for _ in range(num_blocks)])
        self.patch_embedding = DynamicPatchEmbedding(input_shape, embed_dim, patch_size)

    def forward(self, x: torch.Tensor, return_attn=False):
        x = self.patch_embedding(x)
        attn_probs = None
        for blk in self.encoder_blocks:
            x, attn_probs = blk(x, return_attn)
        cls_token_output = x[:, 0]
        return (cls_token_output, attn_probs) if return_attn else cls_token_output


class _DINORuntime:
    def __init__(self, model: "DINOViT", device):
        self.model, self.device = model, device
        C, Hh, Ww = model.input_shape
        P, D, K = model.patch_size, model.embed_dim, model.output_dim
        grid = (Hh // P, Ww // P)
        self.D, self.K = D, K
        self.stores, self.bb, self.head = {}, {}, {}
        for who in ("teacher", "student"):
            st = R.FlatStore(model, device, only=lambda n, who=who: n.startswith(who + "_"), forward_only=(who == "teacher"))
            pre = f"{who}_backbone."
            names = dict(weight=pre + "patch_embedding.proj.weight", bias=pre + "patch_embedding.proj.bias",
                         cls=pre + "patch_embedding.cls_token", pos=pre + "patch_embedding.positional_embedding")
            self.stores[who] = st
            self.bb[who] = BackboneRuntime(st, pre, names, model.num_blocks, C, P, grid, D, model.num_heads, model.mlp_dim,
                                           model.dropout_p)
            self.head[who] = HeadRuntime(st, f"{who}_head.", D, K)
        if self.stores["teacher"].numel != self.stores["student"].numel:
            raise L.VitsslError("DINOViT: teacher and student parameter layouts differ")
        self.ws = R.Workspace()
        self.rec = None
        self.save_gen = 0

    def valid_for(self, device):
        return device == self.device and all(s.is_attached() for s in self.stores.values())

    def forward(self, views: List[torch.Tensor], G: int, training: bool, save: bool):
        m = self.model
        for s in self.stores.values():
            s.refresh_weights()
        V = len(views)
        B = views[0].shape[0]
        dev = views[0].device
        glob = torch.cat([R.as_f32(v) for v in views[:G]], dim=0)
        loc = torch.cat([R.as_f32(v) for v in views[G:]], dim=0) if V > G else None
        seed = R.next_seed() if (training and m.dropout_p > 0) else 0
        student = torch.empty(V * B, self.K, dtype=F32, device=dev)
        fg, _ = self.bb["student"].forward(glob, training, seed, save=save, slot="g", dynamic=True)
        self.head["student"].forward(fg, student[:G * B], save=save, slot="g")
        if loc is not None:
            fl, _ = self.bb["student"].forward(loc, training, seed + 1, save=save, slot="l", dynamic=True)
            self.head["student"].forward(fl, student[G * B:], save=save, slot="l")
        teacher = torch.empty(G * B, self.K, dtype=F32, device=dev)
        ft, _ = self.bb["teacher"].forward(glob, training, seed + 2, save=False, slot="t", dynamic=True)
        self.head["teacher"].forward(ft, teacher, save=False, slot="t")
        self.update_center(teacher)
        if save:
            self.save_gen += 1
            self.rec = dict(G=G, V=V, B=B)
        return teacher, student

    def update_center(self, teacher_out):
        """center <- m c + (1-m) mean_rows(teacher_out) (reference: model.py:91-99), with
        the row sum all-reduced across data-parallel ranks."""
        m = self.model
        cs = self.ws.get("center_colsum", (self.K,), F32, teacher_out.device)
        ops.colsum_f32(teacher_out, cs)
        world = 1
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(cs)
            world = dist.get_world_size()
        ops.center_ema(m.center.view(-1), cs, m.center_momentum, 1.0 / (teacher_out.shape[0] * world))

    def backward(self, dstudent_bf16, reducer=None):
        rec = self.rec
        G, V, B = rec["G"], rec["V"], rec["B"]
        head, bb = self.head["student"], self.bb["student"]
        head.begin_backward()
        if V > G:
            dfl = head.backward(dstudent_bf16[G * B:], "l")
            bb.backward(dfl, "l", None)
        dfg = head.backward(dstudent_bf16[:G * B], "g")
        head.finish_backward()
        st = self.stores["student"]
        if reducer is not None:
            names = [n for n in st.names if n.startswith("student_head.")]
            reducer.ready(*st.span(names[0], names[-1]))
        bb.backward(dfg, "g", reducer)
        if reducer is not None:
            names = [n for n in st.names if n.startswith("student_backbone.patch_embedding.")]
            reducer.ready(*st.span(names[0], names[-1]))


class _DINOFn(Function):
    @staticmethod
    def forward(ctx, rt, views, G, training, need, *params):
        teacher, student = rt.forward(list(views), G, training, save=need)
        ctx.rt, ctx.gen = rt, rt.save_gen
        ctx.mark_non_differentiable(teacher)
        return teacher, student

    @staticmethod
    def backward(ctx, _dt, dstudent):
        rt = ctx.rt
        R.check_saved_generation("DINOViT", ctx.gen, rt.save_gen)
        st = rt.stores["student"]
        st.gflat.zero_()
        d = R.as_f32(dstudent)
        db = torch.empty(d.shape, dtype=BF16, device=d.device)
        ops.cast_bf16(d, db)
        rt.backward(db)
        grads = [st.gview(n, p.shape).clone() if p.requires_grad else None for n, p in zip(st.names, st.params)]
        return (None, None, None, None, None, *grads)


class DINOViT(nn.Module):
    def __init__(
        self,
        num_blocks: int,
        input_shape,
        embed_dim: int,
        patch_size: int,
        num_heads: int = 8,
        mlp_dim: int = 3072,
        dropout: float = 0.1,
        output_dim: int = 65536,
        center_momentum: float = 0.9,
    ):
        super().__init__()
        self.center_momentum = center_momentum
        self.teacher_backbone = ViTBackbone(num_blocks, input_shape, embed_dim, patch_size, num_heads, mlp_dim, dropout)
        self.student_backbone = copy.deepcopy(self.teacher_backbone)
        self.teacher_head = DINOHead(embed_dim, output_dim)
        self.student_head = DINOHead(embed_dim, output_dim)
        self.student_head.load_state_dict(self.teacher_head.state_dict())
        for p in self.teacher_backbone.parameters():
            p.requires_grad = False
        for p in self.teacher_head.parameters():
            p.requires_grad = False
        self.register_buffer("center", torch.zeros(1, output_dim))
        self.input_shape = tuple(input_shape)
        self.num_blocks, self.embed_dim, self.patch_size = num_blocks, embed_dim, patch_size
        self.num_heads, self.mlp_dim, self.dropout_p, self.output_dim = num_heads, mlp_dim, float(dropout), output_dim
        self._rt = None

    # ------------------------------------------------------------------ runtime
    def runtime(self, device=None) -> _DINORuntime:
        device = device or self.center.device
        if device.type != "cuda":
            raise L.VitsslError("DINOViT: parameters are on the CPU; move the model to 'cuda' (no CPU fallback)")
        if self._rt is None or not self._rt.valid_for(device):
            L.lib()
            object.__setattr__(self, "_rt", _DINORuntime(self, device))
        return self._rt

    def flat_store(self):
        return self.runtime().stores["student"]

    def trainable_store(self):
        return self.runtime().stores["student"]

    def all_stores(self):
        return list(self.runtime().stores.values())

    # ------------------------------------------------------------------ reference API
    def forward(self, multi_crop_views: List[torch.Tensor], num_global_views: int):
        R.require_gpu(multi_crop_views[0], "DINOViT")
        rt = self.runtime(multi_crop_views[0].device)
        params = rt.stores["student"].params
        need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        teacher, student = _DINOFn.apply(rt, tuple(multi_crop_views), num_global_views, self.training, need, *params)
        return teacher, student

    @torch.no_grad()
    def momentum_update_teacher(self, teacher_momentum):
        """theta_t <- m theta_t + (1-m) theta_s for every parameter (buffers untouched), as
        one kernel over the two flat stores (reference: model.py:126-139)."""
        rt = self.runtime()
        ops.ema(rt.stores["teacher"].flat, rt.stores["student"].flat, float(teacher_momentum))
        rt.stores["teacher"].mark_dirty()

    @torch.no_grad()
    def inference_forward(self, x: torch.Tensor, return_features=False):
        """Teacher backbone CLS features (or teacher head output); switches to eval like
        the reference (model.py:141-155)."""
        self.eval()
        R.require_gpu(x, "DINOViT")
        rt = self.runtime(x.device)
        rt.stores["teacher"].refresh_weights()
        feats, _ = rt.bb["teacher"].forward(x, False, 0, save=False, slot="inf", dynamic=True)
        if return_features:
            return feats
        out = torch.empty(feats.shape[0], rt.K, dtype=F32, device=x.device)
        rt.head["teacher"].forward(feats, out, save=False, slot="inf")
        return out

    # ------------------------------------------------------------------ fused step
    def train_step(self, views: List[torch.Tensor], num_global_views: int, criterion, optimizer, reducer=None,
                   teacher_momentum: float = 0.996) -> torch.Tensor:
        """Full DINO step without autograd: forward (student x2, teacher, centre), fused
        loss + student-logit gradient, backward, gradient all-reduce, flat AdamW, flat EMA
        (reference: utils/trainers/dino_trainer.py:82-105)."""
        R.require_gpu(views[0], "DINOViT.train_step")
        rt = self.runtime(views[0].device)
        st = rt.stores["student"]
        if getattr(self, "_pacer", None) is None:
            object.__setattr__(self, "_pacer", R.StepPacer())
        self._pacer.begin_step()
        with torch.no_grad():
            st.gflat.zero_()
            if reducer is not None:
                reducer.begin()
            teacher, student = rt.forward(list(views), num_global_views, True, save=True)
            G, V, B, K = num_global_views, len(views), views[0].shape[0], rt.K
            dev = student.device
            loss = rt.ws.get("loss", (1,), F32, dev)
            loss.zero_()
            t_ws = rt.ws.get("t_ws", (ops.dino_tws_floats(G, B, K),), F32, dev)
            dstudent = rt.ws.get("dstudent", (V * B, K), BF16, dev)
            ops.dino_loss(teacher, student, self.center.view(-1), t_ws, loss, dstudent, G, V, B, K,
                          float(criterion.teacher_temp), float(criterion.student_temp), 1.0)
            rt.backward(dstudent, reducer)
            gscale = 1.0
            if reducer is not None:
                reducer.finish()
                gscale = reducer.grad_scale
            optimizer.step_flat(gscale)
            self.momentum_update_teacher(teacher_momentum)
            self.last_teacher, self.last_student = teacher, student
            out = loss[0].clone()
            self._pacer.end_step()
            return out
