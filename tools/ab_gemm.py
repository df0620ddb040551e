#!/usr/bin/env python3
"""Interleaved A/B of NT-GEMM builds in ONE process (guide rule: perf deltas come from interleaved
rounds, not from separate invocations).  Every variant is its own shared library (tools/build_variant.sh),
loaded side by side through ctypes.

    LIBS="main,r64,aux2" [SHAPES="N,K;N,K"] [EPIS="bf16,gelu+drop,..."] python tools/ab_gemm.py

`main` = the in-tree library.  Prints the median time per variant and its ratio to the first one.
Developer tool."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
from vitssl_hip import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")


def load(name):
    path = os.path.join(ROOT, "vit-ssl_amd", "vitssl_hip", "libvitssl_hip.so") if name == "main" else os.path.join(
        ROOT, "tools", "build", f"libvitssl_{name}.so")
    lib = C.CDLL(path)
    lib.vitssl_last_error.restype = C.c_char_p
    lib.vitssl_gemm_bf16_nt.argtypes = [C.POINTER(L.Gemm), C.c_void_p]
    lib.vitssl_gemm_bf16_nt.restype = C.c_int
    lib.vitssl_gemm_bf16_tn.argtypes = L.PROTOTYPES["vitssl_gemm_bf16_tn"]
    lib.vitssl_gemm_bf16_tn.restype = C.c_int
    lib.vitssl_gemm_tn_workspace_floats.restype = C.c_int64
    lib.vitssl_gemm_tn_workspace_floats.argtypes = [C.c_int64, C.c_int, C.c_int]
    return lib


def tn_mode(names, libs, M, rounds, iters, st):
    """C[N1,N2] += A[M,N1]^T B[M,N2] (weight gradients); SHAPES="N1,N2;..." """
    shapes = [(768, 768), (2304, 768), (3072, 768), (768, 3072)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in sk.split(",")) for sk in os.environ["SHAPES"].split(";")]
    rb = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)  # noqa: E731
    for (N1, N2) in shapes:
        A, B = rb(M, N1), rb(M, N2)
        Cm = torch.zeros(N1, N2, device=DEV)
        wsn = int(libs[0].vitssl_gemm_tn_workspace_floats(M, N1, N2))
        ws = torch.empty(max(wsn, 1), device=DEV)

        def run(lib):
            rc = lib.vitssl_gemm_bf16_tn(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N1, N2, ws.data_ptr(), ws.numel(), st)
            if rc != 0:
                raise RuntimeError(lib.vitssl_last_error().decode())

        for lib in libs:
            run(lib)
            run(lib)
        torch.cuda.synchronize()
        times = [[] for _ in libs]
        for _ in range(rounds):
            for li, lib in enumerate(libs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(lib)
                e1.record()
                times[li].append((e0, e1))
        torch.cuda.synchronize()
        fl = 2.0 * M * N1 * N2
        line = f"tn {N1}x{N2}x{M}"
        base = None
        for li in range(len(libs)):
            ts = sorted(a.elapsed_time(b) / iters * 1e3 for a, b in times[li])
            m = ts[len(ts) // 2]
            base = base or m
            line += f" | {names[li]} {m:7.1f} us (min {ts[0]:6.1f}) {fl / m / 1e6:6.0f} TF/s x{m / base:.3f}"
        print(line, flush=True)


def main():
    names = os.environ.get("LIBS", "main").split(",")
    libs = [load(n) for n in names]
    M = int(os.environ.get("M", 50176))
    shapes = [(768, 768), (3072, 768), (768, 3072)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in sk.split(",")) for sk in os.environ["SHAPES"].split(";")]
    want = os.environ.get("EPIS", "bf16,resid+drop,gelu+drop,dgelu+drop,f32").split(",")
    rounds, iters = int(os.environ.get("ROUNDS", 9)), int(os.environ.get("ITERS", 3))
    torch.manual_seed(0)
    rb = lambda *s: (torch.randn(*s, device=DEV) * 0.5).to(torch.bfloat16)  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # stream-K workspace per library (each build has its own registration); SK="1,0" = per variant on / off
    sk = [v != "0" for v in os.environ.get("SK", ",".join("1" for _ in names)).split(",")]
    keep = []
    for lib, on in zip(libs, sk + [True] * len(libs)):
        if on and hasattr(lib, "vitssl_set_nt_workspace"):
            lib.vitssl_nt_workspace_bytes.restype = C.c_int64
            ws = torch.empty(int(lib.vitssl_nt_workspace_bytes()), dtype=torch.uint8, device=DEV)
            lib.vitssl_set_nt_workspace.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
            assert lib.vitssl_set_nt_workspace(ws.data_ptr(), ws.numel(), st) == 0
            keep.append(ws)
    print("variants:", " ".join(names), flush=True)
    if os.environ.get("MODE") == "tn":
        return tn_mode(names, libs, M, rounds, iters, st)
    for (N, K) in shapes:
        pad = int(os.environ.get("LDPAD", 0))             # probe builds with -DNT_PROBE_LDPAD=pad read rows K + pad elements apart
        A, B = rb(M, K + pad), rb(N, K + pad)
        if os.environ.get("UNIT_OUT", "1") != "0":          # outputs ~ N(0, 1) like a model's pre-activations (0.5 * 0.5 * sqrt(K) otherwise)
            B = (B.float() * (4.0 / K ** 0.5)).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV) * 0.1
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        o16b = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        o32 = torch.empty(M, N, device=DEV)
        res = torch.randn(M, N, device=DEV)
        cs = torch.zeros(N, device=DEV)
        cases = {
            "bf16": (L.EPI_BF16, dict(out0=o16)),
            "bf16+bias": (L.EPI_BF16, dict(out0=o16, bias=bias)),
            "f32": (L.EPI_F32, dict(out0=o32, bias=bias)),
            "resid+drop": (L.EPI_RESID, dict(out0=o32, aux=res, bias=bias, drop=True)),
            "resid": (L.EPI_RESID, dict(out0=o32, aux=res, bias=bias)),
            "gelu+drop": (L.EPI_GELU, dict(out0=o16, out1=o16b, bias=bias, drop=True)),
            "gelu": (L.EPI_GELU, dict(out0=o16, out1=o16b, bias=bias)),
            "dgelu+drop": (L.EPI_DGELU, dict(out0=o16b, aux=o16, colsum=cs)),
            "dgelu": (L.EPI_DGELU, dict(out0=o16b, aux=o16)),
        }
        for name in want:
            epi, kw = cases[name]
            g = L.Gemm()
            g.A, g.B, g.M, g.N, g.K, g.epilogue = A.data_ptr(), B.data_ptr(), M, N, K, epi
            g.out0 = kw["out0"].data_ptr()
            for f in ("out1", "aux", "bias", "colsum"):
                if f in kw:
                    setattr(g, f, kw[f].data_ptr())
            g.drop = L.Dropout(0.1, 2, 1) if kw.get("drop") else L.Dropout(0.0, 0, 0)

            def run(lib):
                rc = lib.vitssl_gemm_bf16_nt(C.byref(g), st)
                if rc != 0:
                    raise RuntimeError(lib.vitssl_last_error().decode())

            for lib in libs:
                run(lib)
                run(lib)
            torch.cuda.synchronize()
            times = [[] for _ in libs]
            for _ in range(rounds):
                for li, lib in enumerate(libs):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        run(lib)
                    e1.record()
                    times[li].append((e0, e1))
            torch.cuda.synchronize()
            med = []
            for li in range(len(libs)):
                ts = sorted(a.elapsed_time(b) / iters * 1e3 for a, b in times[li])
                med.append((ts[len(ts) // 2], ts[0]))
            fl = 2.0 * M * N * K
            line = f"nt {M}x{N}x{K:5d} {name:11s}"
            if hasattr(libs[0], "vitssl_debug_last_nt_streamk"):
                run(libs[0])
                line += f" sk{libs[0].vitssl_debug_last_nt_streamk()}"
            for li, (m, lo) in enumerate(med):
                line += f" | {names[li]} {m:7.1f} us (min {lo:6.1f}) {fl / m / 1e6:6.0f} TF/s x{m / med[0][0]:.3f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
