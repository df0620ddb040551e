"""DINO head dgrad shape: dX[M,256] = dY[M,65536] . W[65536,256] through the split-K fp32 path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-ssl_amd"))
import torch
from vitssl_hip import ops, _lib as L
dev = torch.device("cuda:0")
for M in (128, 512, 640):
    A = (torch.randn(M, 65536, device=dev) * 0.1).bfloat16(); B = (torch.randn(256, 65536, device=dev) * 0.1).bfloat16()
    out = torch.empty(M, 256, device=dev)
    for _ in range(3): ops.gemm_nt(A, B, out, L.EPI_F32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(A, B, out, L.EPI_F32)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = A.float() @ B.float().t()
    print(f"M={M}: {us:.1f} us  rel err {float((out-ref).norm()/ref.norm()):.2e}")
