"""Calibration only (never part of the product path): what the vendor library reaches on the
engine's GEMM shapes, to judge how much head-room the hand-written tiles have left."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vit-ssl_amd"))
import torch
from vitssl_hip import ops, _lib as L

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

M = 50176
dev = "cuda"
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms_lt = t(lambda: torch.matmul(a, w.t(), out=out))
    ms_me = t(lambda: ops.gemm_nt(a, w, out, L.EPI_BF16))
    fl = 2.0 * M * N * K / 1e9
    print(f"NT M={M} N={N} K={K}: hipblaslt {ms_lt:.3f} ms {fl/ms_lt:.0f} TF/s | ours {ms_me:.3f} ms {fl/ms_me:.0f} TF/s", flush=True)
# wgrad shape: C[N1,N2] = A[M,N1]^T B[M,N2]
for (N1, N2) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    a = torch.randn(M, N1, device=dev).bfloat16(); b = torch.randn(M, N2, device=dev).bfloat16()
    c = torch.zeros(N1, N2, device=dev)
    c16 = torch.empty(N1, N2, device=dev, dtype=torch.bfloat16)
    ms_lt = t(lambda: torch.matmul(a.t(), b, out=c16))
    ms_me = t(lambda: ops.gemm_tn(a, b, c))
    fl = 2.0 * M * N1 * N2 / 1e9
    print(f"TN M={M} N1={N1} N2={N2}: hipblaslt {ms_lt:.3f} ms {fl/ms_lt:.0f} TF/s | ours {ms_me:.3f} ms {fl/ms_me:.0f} TF/s", flush=True)
