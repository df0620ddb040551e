#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one train step (GPU idle at start)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vit_core.ssl.simmim import SimMIMViT
from vitssl_hip.optim import FusedAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
import os as _os
_cfg = {"vit_b": (768, 12, 3072), "vit_s": (384, 6, 1536)}[_os.environ.get("MODEL", "vit_b")]
m = SimMIMViT(12, (3, 224, 224), _cfg[0], 16, _cfg[1], _cfg[2], 0.1, 0.6).to(dev).train()
opt = FusedAdamW(m.flat_store(), lr=1e-4, weight_decay=1e-3)
x = torch.rand(256, 3, 224, 224, device=dev)
for _ in range(3):
    m.train_step(x, opt)
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    t0 = time.perf_counter()
    m.train_step(x, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("host enqueue ms/step:", [round(h, 2) for h in host], " total ms/step:", [round(t, 2) for t in total])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); m.train_step(x, opt); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# ---- same loop without per-step sync (the host runs ahead, like bench.py)
for label, xin in (("device rand", x), ("cpu-generator rand", torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(42)).to(dev))):
    for _ in range(3):
        m.train_step(xin, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.train_step(xin, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"no-sync loop [{label}]: enqueue {(t1-t0)/20*1e3:.2f} ms/step, total {(t2-t0)/20*1e3:.2f} ms/step")
