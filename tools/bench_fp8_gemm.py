#!/usr/bin/env python3
"""Interleaved timing of the forward Linear GEMMs with bf16 and with e4m3 operands (same shapes, same epilogues).

    [M=25088] [SHAPES="N,K;N,K"] python tools/bench_fp8_gemm.py

Developer tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
from vitssl_hip import _lib as L, ops  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    M = int(os.environ.get("M", 25088))
    shapes = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in sk.split(",")) for sk in os.environ["SHAPES"].split(";")]
    rounds, iters = int(os.environ.get("ROUNDS", 9)), int(os.environ.get("ITERS", 3))
    torch.manual_seed(0)
    for (N, K) in shapes:
        A16 = (torch.randn(M, K, device=DEV)).to(torch.bfloat16)
        B16 = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        A8 = torch.empty(M, K, dtype=ops.FP8, device=DEV)
        ops.quantize_fp8(A16, A8)
        B8 = torch.empty(N, K, dtype=ops.FP8, device=DEV)
        ops.quantize_fp8((B16.float() * 64).to(torch.bfloat16), B8)
        alpha = torch.tensor([1.0 / 64], device=DEV)
        bias = torch.randn(N, device=DEV)
        o16, o16b = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        o8 = torch.empty(M, N, dtype=ops.FP8, device=DEV)
        o32, res = torch.empty(M, N, device=DEV), torch.randn(M, N, device=DEV)
        drop = ops.make_dropout(0.1, 1, 2)
        cases = {
            "bf16": (lambda: ops.gemm_nt(A16, B16, o16, L.EPI_BF16),
                     lambda: ops.gemm_fp8_nt(A8, B8, o16, L.EPI_BF16, alpha=alpha)),
            "resid+drop": (lambda: ops.gemm_nt(A16, B16, o32, L.EPI_RESID, bias=bias, aux=res, drop=drop),
                           lambda: ops.gemm_fp8_nt(A8, B8, o32, L.EPI_RESID, alpha=alpha, bias=bias, aux=res, drop=drop)),
            "gelu+drop": (lambda: ops.gemm_nt(A16, B16, o16, L.EPI_GELU, bias=bias, out1=o16b, drop=drop),
                          lambda: ops.gemm_fp8_nt(A8, B8, o16, L.EPI_GELU, alpha=alpha, bias=bias, out1=o16b, out_fp8=o8, drop=drop)),
        }
        gp = (torch.rand(M, N, device=DEV)).to(torch.bfloat16)
        cs = torch.zeros(N, device=DEV)
        qs, qa, a2 = torch.tensor([4.0], device=DEV), torch.zeros(1, device=DEV), torch.tensor([0.5], device=DEV)
        cases["dgelu+colsum"] = (lambda: ops.gemm_nt(A16, B16, o16, L.EPI_DGELU, aux=gp, colsum=cs),
                                 lambda: ops.gemm_fp8_nt(A8, B8, o16, L.EPI_DGELU, alpha=alpha, alpha2=a2, aux=gp, colsum=cs,
                                                         out_fp8=o8, out_scale=qs, out_amax=qa))
        cases["dgelu (fp8: no image)"] = (lambda: ops.gemm_nt(A16, B16, o16, L.EPI_DGELU, aux=gp, colsum=cs),
                                          lambda: ops.gemm_fp8_nt(A8, B8, o16, L.EPI_DGELU, alpha=alpha, alpha2=a2, aux=gp, colsum=cs))
        cases["dgelu (fp8: image, no amax)"] = (lambda: ops.gemm_nt(A16, B16, o16, L.EPI_DGELU, aux=gp, colsum=cs),
                                                lambda: ops.gemm_fp8_nt(A8, B8, o16, L.EPI_DGELU, alpha=alpha, alpha2=a2, aux=gp, colsum=cs,
                                                                        out_fp8=o8, out_scale=qs))
        only = os.environ.get("CASES")
        for name, fns in cases.items():
            if only and not any(name.startswith(c) for c in only.split(",")):
                continue
            for f in fns:
                f(); f()
            torch.cuda.synchronize()
            times = [[], []]
            for _ in range(rounds):
                for li, f in enumerate(fns):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        f()
                    e1.record()
                    times[li].append((e0, e1))
            torch.cuda.synchronize()
            fl = 2.0 * M * N * K
            med = []
            for li in range(2):
                ts = sorted(a.elapsed_time(b) / iters * 1e3 for a, b in times[li])
                med.append(ts[len(ts) // 2])
            print(f"nt {M}x{N}x{K:5d} {name:28s} | bf16 {med[0]:7.1f} us {fl / med[0] / 1e6:6.0f} TF/s | fp8 {med[1]:7.1f} us "
                  f"{fl / med[1] / 1e6:6.0f} TF/s | x{med[0] / med[1]:.2f}", flush=True)


if __name__ == "__main__":
    main()
