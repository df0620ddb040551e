#!/usr/bin/env python3
"""Launch the attention kernels a few times (for rocprofv3 --pmc runs). args: fwd|bwd"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vitssl_hip import ops
dev = torch.device("cuda:0")
Bn, N, H, dh = 256, 196, 12, 64
torch.manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv = rb(Bn * N, 3 * H * dh); out = torch.empty(Bn * N, H * dh, dtype=torch.bfloat16, device=dev)
dout = rb(Bn * N, H * dh); lse = torch.empty(Bn, H, N, device=dev); dqkv = torch.empty_like(qkv); delta = torch.empty(Bn, H, N, device=dev)
for _ in range(5):
    ops.attn_fwd(qkv, out, lse, Bn, N, H, dh)
    if sys.argv[1] == "bwd":
        ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, Bn, N, H, dh)
torch.cuda.synchronize()
