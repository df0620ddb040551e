#!/usr/bin/env python3
"""Diagnostic: where does a weight-gradient workgroup spend a phase of its K loop?  `--build` compiles gemm_tn.hip with
-DVITSSL_TN_STAMPS into tools/build/libtn_stamps.so (CPU box); on the GPU it runs one launch per shape and prints, for the median
workgroup and each wave group, the average time per K-tile of: LOAD issue (24 transposed reads + 4 DMA), vmcnt wait, barrier,
lgkmcnt wait, 32 MFMAs, barrier (two phases per K-tile, summed).  Developer tool."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vit-ssl_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "build")
LIB = os.path.join(OUT, "libtn_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc, flags = "/opt/rocm/bin/hipcc", ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"]
    objs = []
    for src, extra in (("gemm_tn.hip", ["-DVITSSL_TN_STAMPS"] + sys.argv[2:]), ("gemm_nt.hip", []), ("error.cpp", [])):   # (gemm_nt.hip: the CU-count helpers)
        obj = os.path.join(OUT, src + ".tstamps.o")
        subprocess.run([hipcc] + flags + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    print("built", LIB)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        return build()
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
    from vitssl_hip import _lib as L
    lib = C.CDLL(LIB)
    lib.vitssl_last_error.restype = C.c_char_p
    lib.vitssl_gemm_bf16_tn.argtypes = L.PROTOTYPES["vitssl_gemm_bf16_tn"]
    lib.vitssl_gemm_bf16_tn.restype = C.c_int
    lib.vitssl_gemm_tn_workspace_floats.restype = C.c_int64
    lib.vitssl_gemm_tn_workspace_floats.argtypes = [C.c_int64, C.c_int, C.c_int]
    lib.vitssl_debug_set_tn_stamps.argtypes = [C.c_void_p]
    dev = torch.device("cuda:0")
    M = 50176
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device=dev)
    names = ["LOAD issue", "vmcnt wait", "barrier", "lgkm wait", "MFMA issue", "barrier"]
    for (N1, N2) in [(3072, 768), (768, 3072), (2304, 768), (768, 768)]:
        A = (torch.randn(M, N1, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(M, N2, device=dev) * 0.5).to(torch.bfloat16)
        Cm = torch.zeros(N1, N2, device=dev)
        ws = torch.empty(int(lib.vitssl_gemm_tn_workspace_floats(M, N1, N2)), device=dev)
        import time
        lib.vitssl_debug_set_tn_stamps(None)
        t_end = time.time() + float(os.environ.get("WARM_S", 2.0))   # DVFS: the clock the chip holds shows after ~2 s of back-to-back launches
        while time.time() < t_end:
            for _ in range(50):
                lib.vitssl_gemm_bf16_tn(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N1, N2, ws.data_ptr(), ws.numel(), st)
            torch.cuda.synchronize()
        for arm in (0, 1, 1):
            stamps.zero_()
            if lib.vitssl_debug_set_tn_stamps(stamps.data_ptr() if arm else None) != 0:
                raise RuntimeError("cannot set the stamp pointer")
            rc = lib.vitssl_gemm_bf16_tn(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N1, N2, ws.data_ptr(), ws.numel(), st)
            if rc != 0:
                raise RuntimeError(lib.vitssl_last_error().decode())
            torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(256, 2, 8)
        live = s[:, 0, 6] > 0
        print(f"--- {N1} x {N2}, M = {M}: {int(live.sum())} workgroups, {int(np.median(s[live, 0, 6]))} K-tiles each; us per K-tile (median workgroup)")
        for g in (0, 1):
            per = np.median(s[live, g, :6] / s[live, g, 6:7], axis=0) / 100.0
            ghz = np.median(s[live, g, 7] / np.maximum(s[live, g, :6].sum(axis=1), 1)) * 0.1
            print(f"  waves {4 * g}-{4 * g + 3}: " + "  ".join(f"{n} {v:.3f}" for n, v in zip(names, per)) + f"   total {per.sum():.3f}"
                  f"   in-kernel clock {ghz:.2f} GHz")
    lib.vitssl_debug_set_tn_stamps(None)


if __name__ == "__main__":
    main()
