#!/usr/bin/env python3
"""Launch ONE e4m3-operand GEMM shape a few times (for rocprofv3 --pmc runs).  args: nt|tn N K [epi]   (M = 25088:
ViT-L batch 128; nt epilogues 0 = bf16 store, 2 = GELU + dropout + g' + e4m3 image, 3 = residual + dropout,
4 = dGELU + column sums + scaled e4m3 image)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vitssl_hip import ops
dev = torch.device("cuda:0")
kind, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = int(os.environ.get("M", 25088))
torch.manual_seed(0)
q8 = lambda *s: torch.randn(*s, device=dev).clamp(-400, 400).to(torch.bfloat16)  # noqa: E731


def img(t):
    y = torch.empty(t.shape, dtype=ops.FP8, device=dev)
    ops.quantize_fp8(t, y)
    return y


alpha = torch.tensor([1.0 / 64], device=dev)
if kind == "nt":
    A, B = img(q8(M, K)), img(q8(N, K))
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out32 = torch.empty(M, N, device=dev)
    o8 = torch.empty(M, N, dtype=ops.FP8, device=dev)
    gp = torch.rand(M, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    cs = torch.zeros(N, device=dev)
    drop = ops.make_dropout(0.1, 1, 2)
    qs, qa = torch.tensor([4.0], device=dev), torch.zeros(1, device=dev)
    for _ in range(6):
        if epi == 0:
            ops.gemm_fp8_nt(A, B, out16, 0, alpha=alpha)
        elif epi == 2:
            ops.gemm_fp8_nt(A, B, out16, 2, alpha=alpha, bias=bias, out_fp8=o8, drop=drop)
        elif epi == 3:
            ops.gemm_fp8_nt(A, B, out32, 3, alpha=alpha, bias=bias, aux=res, drop=drop)
        else:
            ops.gemm_fp8_nt(A, B, None, 4, alpha=alpha, aux=gp, colsum=cs, out_fp8=o8, out_scale=qs, out_amax=qa)
else:
    A, B = img(q8(M, N)), img(q8(M, K))
    C = torch.zeros(N, K, device=dev)
    for _ in range(6):
        ops.gemm_fp8_tn(A, B, C, alpha2=alpha)
torch.cuda.synchronize()
