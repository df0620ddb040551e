"""AdamW / EMA launch time against the flat-buffer length (DINO's student store is 109 M parameters and its AdamW ran at 3.8 TB/s
in the step's trace where ViT-B's 86 M run at 5.9).  Developer probe."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vit-ssl_amd"))
import torch
from vitssl_hip import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n in (22_000_000, 86_400_000, 108_900_000, 108_900_000 + 4096, 128 * 1024 * 1024, 162_000_000, 304_000_000):
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    t2 = torch.randn(n, device=dev)
    us = t(lambda: ops.adamw(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 0.05, 3))
    ue = t(lambda: ops.ema(t2, p, 0.996))
    print(f"n = {n / 1e6:7.1f} M: adamw {us:7.1f} us {28 * n / us / 1e6:5.2f} TB/s | ema {ue:7.1f} us {12 * n / ue / 1e6:5.2f} TB/s", flush=True)
    del p, g, m, v, t2
