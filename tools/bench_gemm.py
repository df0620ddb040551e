#!/usr/bin/env python3
"""Micro-benchmark of the GEMM / attention kernels at the ViT-B/16 SimMIM shapes (random
data, HIP-event timing on the launch stream, interleaved rounds).  Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
from vitssl_hip import _lib as L, ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    M = int(os.environ.get("M", 50176))
    torch.manual_seed(0)
    rb = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)  # noqa: E731
    which = sys.argv[1:] or ["nt", "tn", "attn"]
    if "nt" in which:
        shapes = [(768, 768), (3072, 768), (768, 3072)]
        if os.environ.get("EXTRA"):
            shapes = [(2304, 768), (768, 2304)]
        if os.environ.get("SHAPES"):            # "N,K;N,K"
            shapes = [tuple(int(v) for v in sk.split(",")) for sk in os.environ["SHAPES"].split(";")]
        for (N, K) in shapes:
            A, B = rb(M, K), rb(N, K)
            bias = torch.randn(N, device=DEV)
            o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            o16b = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            o32 = torch.empty(M, N, device=DEV)
            res = torch.randn(M, N, device=DEV)
            cs = torch.zeros(N, device=DEV)
            drop = ops.make_dropout(0.1, 1, 2)
            fl = 2.0 * M * N * K
            rows = [("bf16", lambda: ops.gemm_nt(A, B, o16, L.EPI_BF16)),
                    ("resid+drop", lambda: ops.gemm_nt(A, B, o32, L.EPI_RESID, bias=bias, aux=res, drop=drop)),
                    ("gelu+drop", lambda: ops.gemm_nt(A, B, o16, L.EPI_GELU, bias=bias, out1=o16b, drop=drop)),
                    ("dgelu+drop", lambda: ops.gemm_nt(A, B, o16b, L.EPI_DGELU, aux=o16, colsum=cs, drop=drop)),
                    ("resid", lambda: ops.gemm_nt(A, B, o32, L.EPI_RESID, bias=bias, aux=res)),
                    ("gelu", lambda: ops.gemm_nt(A, B, o16, L.EPI_GELU, bias=bias, out1=o16b)),
                    ("dgelu", lambda: ops.gemm_nt(A, B, o16b, L.EPI_DGELU, aux=o16)),
                    ("f32", lambda: ops.gemm_nt(A, B, o32, L.EPI_F32, bias=bias))]
            for name, fn in rows:
                ms = timeit(fn)
                print(f"nt {M}x{N}x{K:5d} {name:11s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
    if "tn" in which:
        for (N1, N2) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
            A, B = rb(M, N1), rb(M, N2)
            C = torch.zeros(N1, N2, device=DEV)
            ms = timeit(lambda: ops.gemm_tn(A, B, C))
            print(f"tn {N1}x{N2}x{M} {ms*1e3:8.1f} us  {2.0*M*N1*N2/ms/1e9:7.1f} TF/s", flush=True)
    if "attn" in which:
        Bn, N, H, dh = 256, 196, 12, 64
        qkv = rb(Bn * N, 3 * H * dh)
        out = torch.empty(Bn * N, H * dh, dtype=torch.bfloat16, device=DEV)
        dout = rb(Bn * N, H * dh)
        lse = torch.empty(Bn, H, N, device=DEV)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(Bn, H, N, device=DEV)
        f = 4.0 * Bn * H * N * N * dh
        ms = timeit(lambda: ops.attn_fwd(qkv, out, lse, Bn, N, H, dh))
        print(f"attn fwd {ms*1e3:8.1f} us  {f/ms/1e9:7.1f} TF/s")
        ms = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, Bn, N, H, dh))
        print(f"attn bwd {ms*1e3:8.1f} us  {2*f/ms/1e9:7.1f} TF/s (8BHN^2d convention)")


if __name__ == "__main__":
    main()
