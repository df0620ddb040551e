"""Triage of a fuzz finding: supervised ViT (B, img, patch, D, H, F, classes) -- per-parameter gradient error against both oracle modes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import rel_l2
from oracle import vit_oracle as O
from vit_core import ViT
B, img, patch, D, H, F, classes = [int(v) for v in sys.argv[1:8]]
dev = torch.device("cuda:0")
torch.manual_seed(B * 1000 + img + classes)
model = ViT(num_classes=classes, num_blocks=2, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F, dropout=0.0)
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.to(dev).train()
x = torch.rand(B, 3, img, img)
labels = torch.randint(0, classes, (B,))
logits, attn = model(x.to(dev), return_attn=True)
torch.nn.CrossEntropyLoss()(logits, labels.to(dev)).backward()
print("logits", logits.detach().flatten().tolist())
for emu in ("bf16", None):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, pr = O.vit_forward(leaves, x, patch, H, emu=emu, return_attn=True)
    O.cross_entropy_mean(lo, labels).backward()
    print("oracle", emu, "logits rel", rel_l2(logits, lo), lo.detach().flatten().tolist())
    for k, p in model.named_parameters():
        r = rel_l2(p.grad, leaves[k].grad)
        print(f"   {k:55s} {r:.4f}  |ref| {float(leaves[k].grad.norm()):.3e}  |got| {float(p.grad.norm()):.3e}")
