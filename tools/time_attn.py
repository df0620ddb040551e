#!/usr/bin/env python3
"""Median time of the attention forward / backward launches at one geometry (developer tool; compare builds or
VITSSL_ATTN_* knobs by running it in alternation).   B H N as arguments, default 256 12 196."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
from vitssl_hip import ops  # noqa: E402

B, H, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (256, 12, 196)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
qkv = torch.randn(B * N, 3 * H * 64, device=dev).to(torch.bfloat16)
out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, N, device=dev)
dout = torch.randn(B * N, H * 64, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, N, device=dev)


def timeit(fn, rounds=15, iters=4):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


f = timeit(lambda: ops.attn_fwd(qkv, out, lse, B, N, H, 64))
b = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, 64))
fl = 4.0 * B * H * N * N * 64
print(f"B{B} H{H} N{N}: fwd {f:7.1f} us {fl / f / 1e6:6.1f} TF/s | bwd {b:7.1f} us {2.5 * fl / b / 1e6:6.1f} TF/s   "
      f"[{os.environ.get('VITSSL_ATTN_FWD_TRIM', '-')}]", flush=True)
