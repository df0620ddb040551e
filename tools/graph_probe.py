#!/usr/bin/env python3
"""Measurement only: eager loop vs the same step replayed from a captured HIP graph
(fixed mask, fixed dropout seed) -> how much of the step is host-induced GPU idle?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vit_core.ssl.simmim import SimMIMViT
from vit_core.ssl.simmim.masking import draw_mask
from vitssl_hip.optim import FusedAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = SimMIMViT(12, (3, 224, 224), 768, 16, 12, 3072, 0.1, 0.6).to(dev).train()
opt = FusedAdamW(m.flat_store(), lr=1e-4, weight_decay=1e-3)
x = torch.rand(256, 3, 224, 224, device=dev)
rt = m.runtime()
prep = rt.prepare_mask(draw_mask(256, 196, 0.6), dev)
def loop(n, **kw):
    for _ in range(3): m.train_step(x, opt, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.train_step(x, opt, **kw)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager, fresh masks     : %.2f ms/step" % loop(20))
print("eager, prepared mask   : %.2f ms/step" % loop(20, prepared=prep))
m._pacer.depth = 10**6
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): m.train_step(x, opt, prepared=prep)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        loss = m.train_step(x, opt, prepared=prep)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("graph replay           : %.2f ms/step  (loss %.4f)" % ((time.perf_counter() - t0) / 20 * 1e3, float(loss)))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
