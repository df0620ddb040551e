#!/usr/bin/env python3
"""Interleaved timing of the weight-gradient GEMM with bf16 and with e4m3 operands.   [M=25088] [SHAPES="N1,N2;..."]
Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
from vitssl_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")
M = int(os.environ.get("M", 25088))
shapes = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in sk.split(",")) for sk in os.environ["SHAPES"].split(";")]
for (N1, N2) in shapes:
    A16 = torch.randn(M, N1, device=DEV).to(torch.bfloat16)
    B16 = torch.randn(M, N2, device=DEV).to(torch.bfloat16)
    A8, B8 = torch.empty(M, N1, dtype=ops.FP8, device=DEV), torch.empty(M, N2, dtype=ops.FP8, device=DEV)
    ops.quantize_fp8(A16, A8)
    ops.quantize_fp8(B16, B8)
    C = torch.zeros(N1, N2, device=DEV)
    fns = (lambda: ops.gemm_tn(A16, B16, C), lambda: ops.gemm_fp8_tn(A8, B8, C))
    for f in fns:
        f(); f()
    torch.cuda.synchronize()
    times = [[], []]
    for _ in range(9):
        for li, f in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            times[li].append((e0, e1))
    torch.cuda.synchronize()
    med = []
    for li in range(2):
        ts = sorted(a.elapsed_time(b) / 3 * 1e3 for a, b in times[li])
        med.append(ts[len(ts) // 2])
    fl = 2.0 * M * N1 * N2
    print(f"tn {N1}x{N2}x{M} | bf16 {med[0]:7.1f} us {fl / med[0] / 1e6:6.0f} TF/s | fp8 {med[1]:7.1f} us {fl / med[1] / 1e6:6.0f} TF/s | x{med[0] / med[1]:.2f}", flush=True)
