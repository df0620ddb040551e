#!/usr/bin/env python3
"""DINO ViT-B/16 step timing (BASELINE configs[3] shapes: 2 x 224^2 global + 8 x 96^2 local
crops, K = 65536, EMA m = 0.996, tau_s 0.1, tau_t 0.04, bf16 GEMMs, AdamW), synthetic data.
Prints image-sets/s and the SURVEY section-8(d) algorithmic FLOP rate (437.8 GF per set)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vit_core.ssl.dino import DINOViT
from vit_core.ssl.dino.loss import DINOLoss
from vitssl_hip.optim import FusedAdamW
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--dropout", type=float, default=0.1)
ap.add_argument("--raw", action="store_true", help="start every step from a uint8 [B,96,96,3] batch: views made by GPUMultiCrop")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(42)
m = DINOViT(12, (3, 224, 224), 768, 16, 12, 3072, a.dropout, 65536, 0.9).to(dev).train()
opt = FusedAdamW(m.trainable_store(), lr=1e-4, weight_decay=1e-3)
crit = DINOLoss(0.04, 0.1)
B = a.batch
from vit_core._runtime import limit_host_threads
limit_host_threads()
views = [torch.rand(B, 3, 224, 224, device=dev) for _ in range(2)] + [torch.rand(B, 3, 96, 96, device=dev) for _ in range(8)]
mc = raw = None
if a.raw:
    from data import GPUMultiCrop, ViewSpec
    mc = GPUMultiCrop(ViewSpec(size=224, scale=(0.5, 1.0), gray_p=0.2), ViewSpec(size=96, scale=(0.08, 0.4)), 10, 2)
    raw = torch.randint(0, 256, (B, 96, 96, 3), dtype=torch.uint8, device=dev)
step_views = (lambda: mc(raw)) if a.raw else (lambda: views)
for _ in range(a.warmup):
    loss = m.train_step(step_views(), 2, crit, opt, None, 0.996)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = m.train_step(step_views(), 2, crit, opt, None, 0.996)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
# one instrumented step: the HBM-bound kernels of the DINO path against the 8 TB/s spec
from vitssl_hip import ops
ops.PROFILE = []
m.train_step(step_views(), 2, crit, opt, None, 0.996)
torch.cuda.synchronize()
recs, ops.PROFILE = ops.PROFILE, None
hbm, fam = {}, {}
for label, flops, e0, e1, nbytes in recs:
    if nbytes:
        h = hbm.setdefault(label, [0.0, 0.0, 0])
        h[0] += nbytes; h[1] += e0.elapsed_time(e1); h[2] += 1
    elif flops:
        f = fam.setdefault(label.split("[")[0].split(" ")[0], [0.0, 0.0, 0])
        f[0] += flops; f[1] += e0.elapsed_time(e1); f[2] += 1
fam = {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1), "ms_per_step": round(v[1], 3), "launches": v[2], "peak": 2500.0} for k, v in fam.items()}
dom = max(fam.items(), key=lambda kv: kv[1]["ms_per_step"]) if fam else None
roofline = None if dom is None else {"kernel": dom[0], "bound": "mfma", "achieved": dom[1]["tflops"], "peak": 2500.0, "unit": "TFLOP/s",
                                     "frac": round(dom[1]["tflops"] / 2500.0, 4), "families": fam}
hbm = {k: {"achieved_tbs": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "peak_tbs": 8.0, "frac": round(v[0] / (v[1] * 1e-3) / 8e12, 3),
           "ms_per_step": round(v[1], 3), "launches": v[2]} for k, v in hbm.items()}
print(json.dumps({"workload": f"ViT-B/16 DINO 2x224+8x96 K=65536 batch {B}" + (" from raw uint8 images (GPU multi-crop)" if a.raw else ""), "ms_per_step": round(dt * 1e3, 2),
                  "image_sets_per_s": round(B / dt, 1), "alg_tflops": round(437.8e9 * B / dt / 1e12, 1),
                  "mfma_util": round(437.8e9 * B / dt / 2.5e15, 4), "loss": round(float(loss), 5), "roofline": roofline, "hbm": hbm}))
