#!/usr/bin/env python3
"""Interleaved A/B of LayerNorm builds (see ab_gemm.py).  LIBS="main,lnA,..." python tools/ab_ln.py"""
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
    for fn in ("vitssl_layernorm_fwd", "vitssl_layernorm_bwd"):
        getattr(lib, fn).argtypes = L.PROTOTYPES[fn]
        getattr(lib, fn).restype = C.c_int
    return lib


def main():
    names = os.environ.get("LIBS", "main").split(",")
    libs = [load(n) for n in names]
    M, D = int(os.environ.get("M", 50176)), int(os.environ.get("D", 768))
    rounds, iters = int(os.environ.get("ROUNDS", 9)), int(os.environ.get("ITERS", 5))
    x = torch.randn(M, D, device=DEV)
    dy = (torch.randn(M, D, device=DEV) * 0.5).bfloat16()
    gres, gout = torch.randn(M, D, device=DEV), torch.empty(M, D, device=DEV)
    gm, y = torch.empty(M, D, device=DEV, dtype=torch.bfloat16), torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    dgam, dbet, cs = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    mo, ro = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    nodrop, drop = L.Dropout(0.0, 0, 0), L.Dropout(0.1, 2, 1)
    cases = {
        "ln_fwd": (6, lambda lib: lib.vitssl_layernorm_fwd(p(x), p(gamma), p(beta), p(y), p(mo), p(ro), M, D, 1e-5, st)),
        "ln_bwd": (16, lambda lib: lib.vitssl_layernorm_bwd(p(dy), p(x), p(mo), p(ro), p(gamma), p(gres), p(gout), p(gm), p(dgam), p(dbet), None, nodrop, M, D, st)),
        "ln_bwd+cs+drop": (16, lambda lib: lib.vitssl_layernorm_bwd(p(dy), p(x), p(mo), p(ro), p(gamma), p(gres), p(gout), p(gm), p(dgam), p(dbet), p(cs), drop, M, D, st)),
    }
    libs[0].vitssl_layernorm_fwd(p(x), p(gamma), p(beta), p(y), p(mo), p(ro), M, D, 1e-5, st)
    print("variants:", " ".join(names), flush=True)
    for name, (bpe, fn) in cases.items():
        for lib in libs:
            assert fn(lib) == 0, lib.vitssl_last_error()
        torch.cuda.synchronize()
        times = [[] for _ in libs]
        for _ in range(rounds):
            for li, lib in enumerate(libs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    fn(lib)
                e1.record()
                times[li].append((e0, e1))
        torch.cuda.synchronize()
        line = f"{name:16s}"
        base = None
        for li in range(len(libs)):
            ts = sorted(a.elapsed_time(b) / iters * 1e3 for a, b in times[li])
            m = ts[len(ts) // 2]
            base = base or m
            line += f" | {names[li]} {m:7.1f} us {M * D * bpe / m / 1e6:5.2f} TB/s x{m / base:.3f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
