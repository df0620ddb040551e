"""SimMIMViT on the MI355X HIP engine (reference: vit_core/ssl/simmim/model.py:8-93).

Same constructor, forward signature, returns and state_dict keys as the reference.
Per step: patch gather (bf16) -> projection GEMM whose epilogue substitutes the mask
token and adds the positional embedding -> L fused encoder blocks -> masked-row gather
-> reconstruction-head GEMM.  Targets are gathered straight from the image in fp32
(bit-exact with ``patches[bool_mask]``).

Two ways to train:
  * reference style: ``pred, tgt = model(x); loss = criterion(pred, tgt); loss.backward()``
    (one torch.autograd.Function around the whole engine), any torch optimizer;
  * fused: ``loss = model.train_step(x, optimizer, reducer)`` -- L1 loss, backward,
    overlapped RCCL gradient all-reduce and flat AdamW with no autograd graph at all
    (what utils.trainers.SimMIMTrainer and bench.py use).
"""
import torch
from torch import nn
from torch.autograd import Function

from ... import _runtime as R
from ..._runtime import BF16, F32, L, ops
from ...encoder_block import EncoderBlock
from .masking import draw_mask, mask_indices_np


class _SimMIMRuntime:
    """Flat store + encoder stack + step workspaces of one SimMIMViT on one device."""

    def __init__(self, model: "SimMIMViT", device):
        self.model = model
        self.device = device
        C, H, W = model.input_shape
        self.P = model.patch_size
        self.C, self.H, self.W = C, H, W
        self.N = (H // self.P) * (W // self.P)
        self.Pd = C * self.P * self.P
        self.D = model.embed_dim
        self.store = R.FlatStore(model, device)
        st = self.store
        prefixes = [f"encoder_blocks.{i}." for i in range(len(model.encoder_blocks))]
        self.stack = R.EncoderStack(st, prefixes, self.D, model.num_heads, model.mlp_dim, model.dropout_p)
        st.register_weight("proj", lambda: st.view("projection.weight", (self.D, self.Pd)), transposed_too=False)
        st.register_weight("head", lambda: st.view("simmim_head.weight", (self.Pd, self.D)))
        self.ws = R.Workspace()
        self.rec = None
        self.save_gen = 0      # id of the forward whose activations `rec` / the stack hold

    def valid_for(self, device) -> bool:
        return device == self.device and self.store.is_attached()

    # ------------------------------------------------------------------ forward
    def embed(self, x, mask_d, tag=""):
        """image -> encoder input tokens fp32 [B*N, D] (also leaves bf16 patches in ws).
        `tag` separates the buffers of forwards that save nothing from those a pending backward
        still needs."""
        st, ws = self.store, self.ws
        B = x.shape[0]
        M = B * self.N
        patches = ws.get(tag + "patches", (M, self.Pd), BF16, x.device)
        ops.patchify_bf16(x, patches, self.P)
        x0 = ws.get(tag + "x0", (M, self.D), F32, x.device)
        ops.gemm_nt(patches, st.w("proj"), x0, L.EPI_EMBED, bias=st.view("projection.bias"),
                    embed=(mask_d, st.view("mask_token") if mask_d is not None else None,
                           st.view("positional_embedding", (self.N, self.D)), self.N, self.N, 0))
        return x0, patches

    def prepare_mask(self, mask_cpu, dev):
        """Host mask -> (idx, inv, mask_u8) device tensors.

        Staging goes through a small RING of persistent pinned host buffers (one
        hipHostMalloc per shape, ever): pinning fresh pages every step (`.pin_memory()`)
        costs milliseconds and serialises with the launch queue on ROCm -- measured 15 ms
        of GPU idle per step.  A slot is reused only after the event recorded behind its
        last host-to-device copies has completed."""
        idx, inv, mflat = mask_indices_np(mask_cpu.contiguous())
        M, Mm = inv.size, idx.size
        key = (M, Mm, str(dev))
        if getattr(self, "_stage_key", None) != key:
            nslots = 4
            self._stage = [dict(idx=torch.empty(Mm, dtype=torch.int32).pin_memory(),
                                inv=torch.empty(M, dtype=torch.int32).pin_memory(),
                                mask=torch.empty(M, dtype=torch.uint8).pin_memory(), ev=None) for _ in range(nslots)]
            self._stage_dev = [dict(idx=torch.empty(Mm, dtype=torch.int32, device=dev),
                                    inv=torch.empty(M, dtype=torch.int32, device=dev),
                                    mask=torch.empty(M, dtype=torch.uint8, device=dev)) for _ in range(nslots)]
            self._stage_pos = 0
            self._stage_key = key
        k = self._stage_pos % len(self._stage)
        self._stage_pos += 1
        slot, d = self._stage[k], self._stage_dev[k]
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        slot["idx"].numpy()[:] = idx          # plain memcpy into the pinned pages (no torch CPU op)
        slot["inv"].numpy()[:] = inv
        slot["mask"].numpy()[:] = mflat
        for name in ("idx", "inv", "mask"):
            d[name].copy_(slot[name], non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record(torch.cuda.current_stream())
        return d["idx"], d["inv"], d["mask"]

    def forward(self, x, training: bool, save: bool, mask_cpu=None, prepared=None):
        st, ws = self.store, self.ws
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.C, self.H, self.W):
            raise L.VitsslError(f"SimMIMViT: expected input [B,{self.C},{self.H},{self.W}], got {tuple(x.shape)}")
        x = R.as_f32(x)
        B = x.shape[0]
        M = B * self.N
        st.refresh_weights()
        dev = x.device
        if prepared is None:
            if mask_cpu is None:
                mask_cpu = draw_mask(B, self.N, self.model.mask_ratio)      # host RNG first (reference order)
            prepared = self.prepare_mask(mask_cpu, dev)
        seed = R.next_seed() if (training and self.stack.p > 0) else 0
        idx_d, inv_d, mask_d = prepared
        Mm = idx_d.numel()

        targets = torch.empty(Mm, self.Pd, dtype=F32, device=dev)
        if Mm == 0:
            # int(N * mask_ratio) == 0 (a single token, or ratio 0): the reference indexes with an all-false mask and returns
            # empty pred / targets (simmim/model.py:56-62); nothing observable depends on the encoder then
            if save:
                self.save_gen += 1
                self.rec = dict(B=B, M=M, Mm=0)
            return torch.empty(0, self.Pd, dtype=F32, device=dev), targets, mask_d.view(B, self.N, 1).bool()
        ops.gather_patches_f32(x, idx_d, targets, self.P)
        tag = "" if save else "tmp."
        x0, patches = self.embed(x, mask_d, tag)
        xL, _ = self.stack.forward(x0, B, self.N, training, seed, save=save, slot="a")
        sel = ws.get(tag + "sel", (Mm, self.D), BF16, dev)
        ops.gather_rows_bf16(xL, idx_d, sel)
        pred = torch.empty(Mm, self.Pd, dtype=F32, device=dev)
        ops.gemm_nt(sel, st.w("head"), pred, L.EPI_F32, bias=st.view("simmim_head.bias"))
        if save:
            self.save_gen += 1
            self.rec = dict(B=B, M=M, Mm=Mm, idx=idx_d, inv=inv_d, mask=mask_d, patches=patches, sel=sel)
        return pred, targets, mask_d.view(B, self.N, 1).bool()

    # ------------------------------------------------------------------ backward
    def backward(self, dpred_bf16, reducer=None):
        """dpred_bf16: bf16 [Mm, Pd].  Accumulates every parameter gradient into the
        store's flat gradient buffer (caller zeroes it)."""
        st, ws, rec = self.store, self.ws, self.rec
        B, M, Mm = rec["B"], rec["M"], rec["Mm"]
        if Mm == 0:                                      # empty prediction: every gradient is zero (the buffer already is)
            if reducer is not None:
                reducer.ready(0, st.gflat.numel())
            return
        dev = dpred_bf16.device
        gv = st.gview
        ops.colsum_bf16(dpred_bf16, gv("simmim_head.bias"))
        ops.gemm_tn(dpred_bf16, rec["sel"], gv("simmim_head.weight", (self.Pd, self.D)))
        dsel = ws.get("dsel", (Mm, self.D), BF16, dev)
        ops.gemm_nt(dpred_bf16, st.w("head.T"), dsel, L.EPI_BF16)
        g = ws.get("g", (M, self.D), F32, dev)
        ops.scatter_rows_f32(dsel, rec["inv"], g)
        if reducer is not None:
            reducer.ready(*st.span("simmim_head.weight", "simmim_head.bias"))
        g = self.stack.backward(g, slot="a", reducer=reducer)
        dproj = ws.get("dproj", (M, self.D), BF16, dev)
        ops.embed_bwd(g, rec["mask"], dproj, gv("positional_embedding", (self.N, self.D)), gv("mask_token"),
                      gv("projection.bias"), None, B, self.N, 0, self.D)
        ops.gemm_tn(dproj, rec["patches"], gv("projection.weight", (self.D, self.Pd)))
        if reducer is not None:
            reducer.ready(*st.span("mask_token", "positional_embedding"))
            reducer.ready(*st.span("projection.weight", "projection.bias"))


class _SimMIMFn(Function):
    @staticmethod
    def forward(ctx, rt, x, training, need, *params):
        pred, targets, mask = rt.forward(x, training, save=need)
        ctx.rt, ctx.gen = rt, rt.save_gen
        ctx.mark_non_differentiable(targets, mask)
        return pred, targets, mask

    @staticmethod
    def backward(ctx, dpred, _dt, _dm):
        rt = ctx.rt
        R.check_saved_generation("SimMIMViT", ctx.gen, rt.save_gen)
        st = rt.store
        st.gflat.zero_()
        dp = R.as_f32(dpred)
        dpb = torch.empty(dp.shape, dtype=BF16, device=dp.device)
        if dp.numel() > 0:
            ops.cast_bf16(dp, dpb)
        rt.backward(dpb)
        grads = [st.gview(n, p.shape).clone() if p.requires_grad else None for n, p in zip(st.names, st.params)]
        return (None, None, None, None, *grads)


class SimMIMViT(nn.Module):
    def __init__(
        self,
        num_blocks: int,
        input_shape,
        embed_dim: int,
        patch_size: int,
        num_heads: int = 8,
        mlp_dim: int = 3072,
        dropout: float = 0.1,
        mask_ratio: float = 0.6,
    ):
        super().__init__()
        self.encoder_blocks = nn.ModuleList(
            [EncoderBlock(embed_dim, num_heads, mlp_dim, dropout) for _ in range(num_blocks)]
        )
        self.unfold = nn.Unfold(kernel_size=(patch_size, patch_size), stride=patch_size)
        self.projection = nn.Linear((input_shape[0] * patch_size * patch_size), embed_dim)
        self.mask_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.positional_embedding = nn.Parameter(torch.rand(1, (input_shape[1] // patch_size) ** 2, embed_dim))
        self.simmim_head = nn.Linear(embed_dim, input_shape[0] * patch_size * patch_size)

        self.mask_ratio = mask_ratio
        self.input_shape = tuple(input_shape)
        self.embed_dim, self.patch_size = embed_dim, patch_size
        self.num_heads, self.mlp_dim, self.dropout_p = num_heads, mlp_dim, float(dropout)
        if input_shape[1] % patch_size != 0 or input_shape[2] % patch_size != 0:
            raise ValueError(
                f"Image dimensions H={input_shape[1]}, W={input_shape[2]} must be divisible by patch_size={patch_size}")
        self._rt = None

    # ------------------------------------------------------------------ runtime
    def runtime(self, device=None) -> _SimMIMRuntime:
        device = device or self.mask_token.device
        if device.type != "cuda":
            raise L.VitsslError("SimMIMViT: parameters are on the CPU; move the model to 'cuda' (no CPU fallback)")
        if self._rt is None or not self._rt.valid_for(device):
            L.lib()
            object.__setattr__(self, "_rt", _SimMIMRuntime(self, device))
        return self._rt

    def flat_store(self):
        return self.runtime().store

    # ------------------------------------------------------------------ reference API
    def forward(self, x: torch.Tensor, return_bool_mask=False):
        R.require_gpu(x, "SimMIMViT")
        rt = self.runtime(x.device)
        need = torch.is_grad_enabled() and any(p.requires_grad for p in rt.store.params)
        pred, targets, mask = _SimMIMFn.apply(rt, x, self.training, need, *rt.store.params)
        if return_bool_mask:
            return pred, targets, mask
        return pred, targets

    @torch.no_grad()
    def inference_forward(self, x: torch.Tensor, return_patch_features=False):
        """Unmasked encode; mean-pooled [B, D] or all tokens [B, N, D]
        (reference: ssl/simmim/model.py:65-93; switches the module to eval like it)."""
        self.eval()
        R.require_gpu(x, "SimMIMViT")
        rt = self.runtime(x.device)
        rt.store.refresh_weights()
        x = R.as_f32(x)
        B = x.shape[0]
        x0, _ = rt.embed(x, None, "inf.")
        xL, _ = rt.stack.forward(x0, B, rt.N, False, 0, save=False, slot="inf")
        feats = xL.view(B, rt.N, rt.D)
        return feats.clone() if return_patch_features else feats.mean(dim=1)

    # ------------------------------------------------------------------ fused step
    def train_step(self, x: torch.Tensor, optimizer, reducer=None, mask_cpu=None, prepared=None) -> torch.Tensor:
        """One full optimisation step (zero_grad -> forward -> L1(mean) -> backward ->
        gradient all-reduce -> AdamW) with no autograd graph; returns the loss as a
        device scalar (no host sync).  Equivalent to utils/trainers/simmim_trainer.py:61-76
        of the reference with criterion nn.L1Loss(mean)."""
        R.require_gpu(x, "SimMIMViT.train_step")
        rt = self.runtime(x.device)
        st = rt.store
        if getattr(self, "_pacer", None) is None:
            object.__setattr__(self, "_pacer", R.StepPacer())
        self._pacer.begin_step()
        with torch.no_grad():
            st.gflat.zero_()
            if reducer is not None:
                reducer.begin()
            pred, targets, _ = rt.forward(x, True, save=True, mask_cpu=mask_cpu, prepared=prepared)
            n = pred.numel()
            loss_sum = rt.ws.get("loss_sum", (1,), F32, x.device)
            loss_sum.zero_()
            if n == 0:
                # no masked token: the reference's L1Loss(mean) of an empty prediction is nan, every gradient zero, and the
                # optimizer still steps (weight decay only)
                loss_sum.fill_(float("nan"))
                n = 1
                rt.backward(pred, reducer)
            else:
                dpb = rt.ws.get("dpred", tuple(pred.shape), BF16, x.device)
                ops.l1_loss(pred, targets, loss_sum, dpb, gscale=1.0 / n)
                rt.backward(dpb, reducer)
            gscale = 1.0
            if reducer is not None:
                reducer.finish()
                gscale = reducer.grad_scale
            optimizer.step_flat(gscale)
            self.last_pred, self.last_targets = pred, targets
            loss = loss_sum[0] / n
            self._pacer.end_step()
            return loss
