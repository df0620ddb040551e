"""Patch-embed (+CLS, +pos) -> encoder stack -> CLS feature, on a flat store.
Shared by ViT (vit_core/vit.py:33-45) and the DINO backbones
(vit_core/ssl/dino/model.py:34-45)."""
import torch

from . import _runtime as R
from ._runtime import BF16, F32, L, ops


class BackboneRuntime:
    def __init__(self, store, prefix: str, embed_names: dict, num_blocks: int, in_ch: int, patch: int, grid, D: int, H: int,
                 Fd: int, p_drop: float, site_base: int = 0):
        """embed_names: {'weight','bias','cls','pos'} -> parameter names (relative to the
        store) of the patch projection, CLS token and positional embedding."""
        self.store, self.prefix = store, prefix
        self.names = embed_names
        self.C, self.P, self.grid = in_ch, patch, tuple(grid)
        self.D = D
        self.Pd = in_ch * patch * patch
        self.T0 = grid[0] * grid[1] + 1
        blocks = [f"{prefix}encoder_blocks.{i}." for i in range(num_blocks)]
        self.stack = R.EncoderStack(store, blocks, D, H, Fd, p_drop, site_base=site_base)
        self.wkey = prefix + "patch_proj"
        store.register_weight(self.wkey, lambda: store.view(embed_names["weight"], (D, self.Pd)), transposed_too=False)
        self.ws = R.Workspace()
        self.rec = {}

    # positional embedding for a gh x gw patch grid (reference: patch_embedding.py:26-48)
    def pos_for(self, gh: int, gw: int, need_grad: bool):
        st = self.store
        pos = st.view(self.names["pos"], (1, self.T0, self.D))
        if gh * gw == self.grid[0] * self.grid[1] and gh == gw:
            return pos[0], None
        # CLS row copied, patch rows resampled on the grid (bicubic, as F.interpolate does it)
        full = torch.empty(gh * gw + 1, self.D, dtype=F32, device=pos.device)
        full[0] = pos[0, 0]
        ops.bicubic_resize_fwd(pos[0, 1:], full[1:], self.grid[0], self.grid[1], gh, gw)
        return full, (gh, gw)

    def forward(self, x, training: bool, seed: int, save: bool, slot: str, return_attn=False, dynamic=False):
        """x: fp32 [B, C, Himg, Wimg] -> (cls features fp32 [B, D], attn probs or None)."""
        st, ws = self.store, self.ws
        x = R.as_f32(x)
        B, C, Hi, Wi = x.shape
        if Hi % self.P != 0 or Wi % self.P != 0:
            raise ValueError(f"Input image dimensions ({Hi}x{Wi}) must be divisible by patch size ({self.P}).")
        gh, gw = Hi // self.P, Wi // self.P
        if not dynamic and (gh, gw) != self.grid:
            raise L.VitsslError(f"patch embedding built for a {self.grid} grid got {gh}x{gw}")
        tokens = gh * gw
        T = tokens + 1
        dev = x.device
        pos, pos_graph = self.pos_for(gh, gw, need_grad=save)
        # a forward that saves nothing (no_grad / eval) must not touch the buffers a pending
        # backward of the same slot still needs
        wtag = slot if save else slot + ".tmp"
        patches = ws.get(f"{wtag}.patches", (B * tokens, self.Pd), BF16, dev)
        ops.patchify_bf16(x, patches, self.P)
        x0 = ws.get(f"{wtag}.x0", (B * T, self.D), F32, dev)
        ops.gemm_nt(patches, st.w(self.wkey), x0, L.EPI_EMBED, bias=st.view(self.names["bias"]),
                    embed=(None, None, pos, tokens, T, 1))
        x0.view(B, T, self.D)[:, 0] = st.view(self.names["cls"]) + pos[0]
        xL, probs = self.stack.forward(x0, B, T, training, seed, save=save, slot=slot, return_attn=return_attn)
        feats = torch.empty(B, self.D, dtype=F32, device=dev)
        ops.gather_cls_f32(xL, feats, B, T, self.D)
        if save:
            self.rec[slot] = dict(B=B, T=T, tokens=tokens, patches=patches, pos_graph=pos_graph)
        return feats, probs

    def backward(self, dfeats, slot: str, reducer=None):
        """dfeats: fp32 [B, D] gradient wrt the CLS features; accumulates parameter grads."""
        st, ws = self.store, self.ws
        rec = self.rec[slot]
        B, T, tokens = rec["B"], rec["T"], rec["tokens"]
        dev = dfeats.device
        gv = st.gview
        g = ws.get(f"{slot}.g", (B * T, self.D), F32, dev)
        ops.scatter_cls_f32(R.as_f32(dfeats), g, B, T, self.D)
        g = self.stack.backward(g, slot=slot, reducer=reducer)
        dproj = ws.get(f"{slot}.dproj", (B * tokens, self.D), BF16, dev)
        if rec["pos_graph"] is None:
            dpos = gv(self.names["pos"], (T, self.D))
        else:
            dpos = torch.zeros(T, self.D, dtype=F32, device=dev)
        ops.embed_bwd(g, None, dproj, dpos, None, gv(self.names["bias"]), gv(self.names["cls"]), B, tokens, 1, self.D)
        if rec["pos_graph"] is not None:                       # resized table: route d(pos) back through the resize
            gh, gw = rec["pos_graph"]
            gpos = gv(self.names["pos"], (self.T0, self.D))
            gpos[0].add_(dpos[0])                               # CLS row: identity
            ops.bicubic_resize_bwd(dpos[1:], gpos[1:], self.grid[0], self.grid[1], gh, gw)
        ops.gemm_tn(dproj, rec["patches"], gv(self.names["weight"], (self.D, self.Pd)))
        # the CLS position also receives the CLS-row gradient through `cls + pos[0]`:
        # embed_bwd already added row 0 of every image to dpos[0] and to dcls.
