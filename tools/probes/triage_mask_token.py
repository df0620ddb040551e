"""Triage of a fuzz finding: SimMIM (B=1, img=16, patch=8, D=128, H=2, F=64): mask_token gradient rel-L2 0.094 vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import rel_l2
from oracle import vit_oracle as O
from vit_core.ssl.simmim import SimMIMViT
from vit_core.ssl.simmim.masking import draw_mask
DEV = torch.device("cuda:0")
B, img, patch, D, H, F = 1, 16, 8, 128, 2, 64
torch.manual_seed(B * 100 + img)
model = SimMIMViT(num_blocks=2, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F, dropout=0.0, mask_ratio=0.6)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.to(DEV).train()
x = torch.rand(B, 3, img, img)
N = (img // patch) ** 2
torch.manual_seed(77)
mask = draw_mask(B, N, 0.6)
torch.manual_seed(77)
pred, tgt, bm = model(x.to(DEV), return_bool_mask=True)
torch.nn.L1Loss()(pred, tgt).backward()
for emu in ("bf16", None):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pe, te = O.simmim_forward(leaves, x, mask, patch, H, emu=emu)
    O.l1_loss_mean(pe, te).backward()
    print("oracle emu =", emu, " pred rel_l2", rel_l2(pred, pe))
    for k, p in model.named_parameters():
        r = rel_l2(p.grad, leaves[k].grad)
        if r > 1e-2 or k in ("mask_token", "positional_embedding"):
            print(f"  {k:60s} {r:.4f}  |ref| {float(leaves[k].grad.norm()):.3e}")
    gp = leaves["positional_embedding"].grad[0]          # [N, D]: d loss / d x0 per position
    m = mask[0]
    parts = gp[m]
    print("  masked positions:", m.nonzero().flatten().tolist(), " |sum| / sum|.| =", float(parts.sum(0).norm() / parts.norm(dim=1).sum()))
    print("  per-position rel_l2 of dx0:", [round(rel_l2(model.positional_embedding.grad[0, i], gp[i]), 4) for i in range(N)])
