#!/usr/bin/env python3
"""Diagnostic: where does an NT-GEMM workgroup spend its time?  Builds gemm_nt.hip with
-DVITSSL_NT_STAMPS into tools/build/libnt_stamps.so (`--build`, on the CPU box: hipcc cross-compiles),
then on the GPU runs one launch per epilogue with the stamp buffer armed and prints, per output tile of
the median workgroup: K-loop time, epilogue issue time, time until the stores are acknowledged
(s_memrealtime, 100 MHz).  The stamped kernel waits for its stores after every tile, so totals are
slower than the product kernel: read the split, not the sum.  Developer tool."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vit-ssl_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "build")
LIB = os.path.join(OUT, "libnt_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc, flags = "/opt/rocm/bin/hipcc", ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"]
    objs = []
    for src, extra in (("gemm_nt.hip", ["-DVITSSL_NT_STAMPS"]), ("error.cpp", [])):
        obj = os.path.join(OUT, src + ".stamps.o")
        subprocess.run([hipcc] + flags + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    print("built", LIB)


def main():
    if "--build" in sys.argv:
        return build()
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
    from vitssl_hip import _lib as L
    lib = C.CDLL(LIB)
    lib.vitssl_last_error.restype = C.c_char_p
    lib.vitssl_gemm_bf16_nt.argtypes = [C.POINTER(L.Gemm), C.c_void_p]
    lib.vitssl_debug_nt_stamps.argtypes = [C.c_void_p]
    dev = torch.device("cuda:0")
    M = int(os.environ.get("M", 50176))
    rb = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    stamps = torch.zeros(2 * 256 * 2 * 16 * 4, dtype=torch.int64, device=dev)
    warm_s = float(os.environ.get("WARM_S", 2.0))     # DVFS: the clock the chip holds shows after ~2 s of back-to-back launches
    for (N, K) in [(3072, 768), (768, 768), (768, 3072)]:
        A, B = rb(M, K), rb(N, K)
        B = (B.float() * (4.0 / K ** 0.5)).to(torch.bfloat16)     # outputs ~ N(0, 1), like a model's pre-activations
        bias = torch.randn(N, device=dev) * 0.1
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        o16b = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        o32 = torch.randn(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        for name, epi, kw in [("bf16", L.EPI_BF16, dict(out0=o16)), ("gelu+drop", L.EPI_GELU, dict(out0=o16, out1=o16b, bias=bias, drop=True)),
                              ("gelu", L.EPI_GELU, dict(out0=o16, out1=o16b, bias=bias)),
                              ("dgelu", L.EPI_DGELU, dict(out0=o16b, aux=o16)), ("resid+drop", L.EPI_RESID, dict(out0=o32, aux=res, bias=bias, drop=True))]:
            g = L.Gemm()
            g.A, g.B, g.M, g.N, g.K, g.epilogue = A.data_ptr(), B.data_ptr(), M, N, K, epi
            g.out0 = kw["out0"].data_ptr()
            g.out1 = kw["out1"].data_ptr() if "out1" in kw else None
            g.aux = kw["aux"].data_ptr() if "aux" in kw else None
            g.bias = kw["bias"].data_ptr() if "bias" in kw else None
            g.drop = L.Dropout(0.1, 2, 1) if kw.get("drop") else L.Dropout(0.0, 0, 0)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.vitssl_debug_nt_stamps(None)
            import time
            t_end = time.time() + warm_s
            while True:
                for _ in range(50):
                    assert lib.vitssl_gemm_bf16_nt(C.byref(g), st) == 0, lib.vitssl_last_error()
                torch.cuda.synchronize()
                if time.time() >= t_end:
                    break
            for _ in range(20):
                assert lib.vitssl_gemm_bf16_nt(C.byref(g), st) == 0, lib.vitssl_last_error()
            stamps.zero_()
            torch.cuda.synchronize()
            lib.vitssl_debug_nt_stamps(C.c_void_p(stamps.data_ptr()))
            assert lib.vitssl_gemm_bf16_nt(C.byref(g), st) == 0, lib.vitssl_last_error()
            torch.cuda.synchronize()
            lib.vitssl_debug_nt_stamps(None)
            raw = stamps.cpu().numpy().reshape(2, 256, 2, 16, 4).astype(np.float64)
            s = raw[0] / 100.0   # us
            ck = raw[1]          # shader clocks
            okc = (raw[0][:, :, 0, 0] > 0) & (raw[0][:, :, 0, 2] > 0)
            loop_ghz = np.median(((ck[:, :, 0, 1] - ck[:, :, 0, 0]) / np.maximum(raw[0][:, :, 0, 1] - raw[0][:, :, 0, 0], 1))[okc]) * 0.1
            epi_ghz = np.median(((ck[:, :, 0, 2] - ck[:, :, 0, 1]) / np.maximum(raw[0][:, :, 0, 2] - raw[0][:, :, 0, 1], 1))[okc]) * 0.1
            t0 = s[s > 0].min()
            print(f"--- N={N} K={K} {name}: kernel span {s.max() - t0:.1f} us (stamped build)")
            print(f"    in-kernel clock (tile 0, median over workgroups): K loop {loop_ghz:.2f} GHz, epilogue {epi_ghz:.2f} GHz")
            for grp in (0, 1):
                rows = []
                for r in range(16):
                    ok = s[:, grp, r, 0] > 0
                    if ok.sum() < 8:
                        break
                    loop = np.median(s[ok, grp, r, 1] - s[ok, grp, r, 0])
                    epi_t = np.median(s[ok, grp, r, 2] - s[ok, grp, r, 1])
                    ack = np.median(s[ok, grp, r, 3] - s[ok, grp, r, 2])
                    start = np.median(s[ok, grp, r, 0] - t0)
                    rows.append(f"r{r}: start {start:6.1f} loop {loop:5.1f} epi {epi_t:5.1f} ack {ack:4.1f} ({int(ok.sum())} wgs)")
                print(f"  waves {4*grp}-{4*grp+3}: " + " | ".join(rows[:4]) + (" ..." if len(rows) > 4 else ""))


if __name__ == "__main__":
    main()
