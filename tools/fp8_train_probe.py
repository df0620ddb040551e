#!/usr/bin/env python3
"""Loss trajectory of 30 fused training steps on one fixed batch, bf16 operands and then fp8 forward operands
(ViT-L/16 SimMIM, batch 128, dropout 0.1, same seeds), plus the fraction of bytes of one e4m3 weight image that
changed between the first and the last step.  Developer tool: python tools/fp8_train_probe.py"""
import sys, os
sys.path.insert(0, "vit-ssl_amd")
import torch
from vitssl_hip import engine
from vitssl_hip.optim import FusedAdamW
from vit_core.ssl.simmim import SimMIMViT
dev = torch.device("cuda:0")
for mode in ("bf16", "fp8"):
    engine.set_linear_operands(mode)
    torch.manual_seed(42)
    m = SimMIMViT(num_blocks=24, input_shape=(3, 224, 224), embed_dim=1024, patch_size=16, num_heads=16, mlp_dim=4096, dropout=0.1, mask_ratio=0.6).to(dev).train()
    st = m.flat_store()
    opt = FusedAdamW(st, lr=1e-4, weight_decay=1e-3)
    x = torch.rand(128, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    torch.manual_seed(5)
    ls = []
    snap = None
    for i in range(30):
        l = float(m.train_step(x, opt))
        ls.append(round(l, 5))
        if mode == "fp8" and i in (0, 29):
            img, al = st.w8("encoder_blocks.0.w1")
            cur = (img.view(torch.uint8).clone(), float(al))
            if snap is not None:
                print("fp8 image bytes changed:", float((cur[0] != snap[0]).float().mean()), "alpha", snap[1], cur[1])
            snap = cur
    print(mode, ls)
