#!/usr/bin/env python3
"""Launch ONE GEMM shape a few times (for rocprofv3 --pmc runs).  args: kind N K [epi]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
import torch
from vitssl_hip import _lib as L, ops
dev = torch.device("cuda:0")
kind, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
M = 50176
torch.manual_seed(0)
rb = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
if kind == "nt":
    A, B = rb(M, K), rb(N, K)
    B = (B.float() * (4.0 / K ** 0.5)).to(torch.bfloat16)      # outputs ~ N(0, 1), like a model's pre-activations
    out = torch.empty(M, N, dtype=torch.bfloat16 if epi in (0, 2, 4) else torch.float32, device=dev)
    out1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux = torch.randn(M, N, device=dev) if epi == 3 else rb(M, N)
    bias = torch.randn(N, device=dev) * 0.1
    drop = ops.make_dropout(0.1, 1, 2) if (epi in (2, 3) and not os.environ.get("NODROP")) else ops.NO_DROP   # as in the model
    cs = torch.zeros(N, device=dev) if epi == 4 else None
    for _ in range(6):
        ops.gemm_nt(A, B, out, epi, bias=bias, aux=aux if epi in (3, 4) else None, out1=out1 if epi == 2 else None, drop=drop, colsum=cs)
else:
    A, B = rb(M, N), rb(M, K)
    C = torch.zeros(N, K, device=dev)
    for _ in range(6):
        ops.gemm_tn(A, B, C)
torch.cuda.synchronize()
