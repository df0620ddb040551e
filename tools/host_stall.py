#!/usr/bin/env python3
"""Which host-side call blocks?  Times every C-ABI call and the other per-step host actions."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vitssl_hip import _lib as L, ops
from vit_core.ssl.simmim import SimMIMViT
from vit_core.ssl.simmim import model as M
from vit_core import _runtime as R
from vitssl_hip.optim import FusedAdamW
dev = torch.device("cuda:0")
cfg = {"vit_b": (768, 12, 3072), "vit_s": (384, 6, 1536)}[os.environ.get("MODEL", "vit_s")]
torch.manual_seed(0)
m = SimMIMViT(12, (3, 224, 224), cfg[0], 16, cfg[1], cfg[2], 0.1, 0.6).to(dev).train()
opt = FusedAdamW(m.flat_store(), lr=1e-4, weight_decay=1e-3)
x = torch.rand(256, 3, 224, 224, device=dev)
slow = []
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = (time.perf_counter() - t0) * 1e3
        if dt > 2.0: slow.append((label or name, round(dt, 1), a[0] if a and isinstance(a[0], str) else ""))
        return r
    setattr(obj, name, g)
wrap(ops, "call"); wrap(L, "call")
for fn in ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "layernorm_fwd", "layernorm_bwd", "cast_transpose_bf16", "adamw"):
    wrap(ops, fn)
wrap(M, "draw_mask"); wrap(torch, "empty", "torch.empty")
rt = m.runtime()
wrap(rt, "prepare_mask"); 
for _ in range(3): m.train_step(x, opt)
wrap(m._pacer, "begin_step"); wrap(m._pacer, "end_step")
wrap(rt.store.gflat, "zero_", "gflat.zero_")
torch.cuda.synchronize()
for i in range(30):
    t0 = time.perf_counter(); m.train_step(x, opt); dt = (time.perf_counter() - t0) * 1e3
    if dt > 15: print(f"step {i}: host {dt:.1f} ms; slow calls: {slow[-6:]}")
    slow.clear()
torch.cuda.synchronize()
print("done")
