#!/usr/bin/env python3
"""Diagnostic: phase split of the attention kernels per workgroup (s_memrealtime, 100 MHz).
`--build` (CPU box) compiles attention.hip with -DVITSSL_ATTN_STAMPS into tools/build/libattn_stamps.so.
fwd stamps: 0 start, 1 K landed + V issued, 2 first query-pair iteration done, 3 end;
bwd stamps: 0 start, 1 prologue done (tiles landed), 2 query sweep + dQ done, 3 dK/dV stored and acknowledged."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vit-ssl_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "build")
LIB = os.path.join(OUT, "libattn_stamps.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    hipcc, flags = "/opt/rocm/bin/hipcc", ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"]
    objs = []
    for src, extra in (("attention.hip", ["-DVITSSL_ATTN_STAMPS"]), ("error.cpp", [])):
        obj = os.path.join(OUT, src + ".astamps.o")
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
    lib.vitssl_attn_fwd.argtypes = L.PROTOTYPES["vitssl_attn_fwd"]
    lib.vitssl_attn_bwd.argtypes = L.PROTOTYPES["vitssl_attn_bwd"]
    lib.vitssl_debug_attn_stamps.argtypes = [C.c_void_p]
    dev = torch.device("cuda:0")
    Bn, N, H, dh = int(os.environ.get("B", 256)), int(os.environ.get("N", 196)), 12, 64
    rb = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)  # noqa: E731
    qkv, dout = rb(Bn * N, 3 * H * dh), rb(Bn * N, H * dh)
    out = torch.empty(Bn * N, H * dh, dtype=torch.bfloat16, device=dev)
    lse, delta = torch.empty(Bn, H, N, device=dev), torch.empty(Bn, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    nwg = Bn * H
    stamps = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    fwd = lambda: lib.vitssl_attn_fwd(p(qkv), p(out), p(lse), None, Bn, N, H, dh, st)  # noqa: E731
    bwd = lambda: lib.vitssl_attn_bwd(p(qkv), p(out), p(dout), p(lse), p(dqkv), p(delta), Bn, N, H, dh, st)  # noqa: E731
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        assert lib.vitssl_debug_attn_stamps(None) == 0
        for _ in range(3):
            assert fn() == 0, lib.vitssl_last_error()
        stamps.zero_()
        torch.cuda.synchronize()
        assert lib.vitssl_debug_attn_stamps(p(stamps)) == 0
        assert fn() == 0, lib.vitssl_last_error()
        torch.cuda.synchronize()
        assert lib.vitssl_debug_attn_stamps(None) == 0
        if name == "fwd" and N > 128 and os.environ.get("VITSSL_ATTN_FWD_PERSIST", "2") != "0":
            # persistent forward: per workgroup, time of wave 0 per segment summed over its items
            seg = stamps.cpu().numpy().reshape(-1, 8)[:256, :6].astype(np.float64)
            seg = seg[seg[:, 5] > 0]
            per = seg[:, :5] / seg[:, 5:6] / 100.0
            labs = ("wait + barrier", "prefetch issue + S", "softmax", "P.V", "stores")
            print(f"--- attn fwd (persistent): {len(seg)} workgroups x {seg[:, 5].mean():.1f} items; wave 0, us per item (median over workgroups):")
            for i, lab in enumerate(labs):
                print(f"    {lab:20s}: {np.median(per[:, i]):6.2f}")
            print(f"    {'item':20s}: {np.median(per.sum(1)):6.2f}")
            continue
        s = stamps.cpu().numpy().reshape(nwg, 4).astype(np.float64) / 100.0
        t0 = s[:, 0].min()
        d = np.diff(s, axis=1)
        span = s[:, 3].max() - t0
        starts = np.sort(s[:, 0] - t0)
        print(f"--- attn {name}: kernel span {span:.1f} us, {nwg} workgroups; per workgroup median (p10..p90) us:")
        for i, lab in enumerate(("phase 0->1", "phase 1->2", "phase 2->3")):
            print(f"    {lab}: {np.median(d[:, i]):6.2f} ({np.percentile(d[:, i], 10):.2f}..{np.percentile(d[:, i], 90):.2f})")
        tot = s[:, 3] - s[:, 0]
        print(f"    total     : {np.median(tot):6.2f} ({np.percentile(tot, 10):.2f}..{np.percentile(tot, 90):.2f});  "
              f"workgroups resident at once ~ {np.sum(tot) / span:.0f}  (start of the 256th / 512th / 768th workgroup: "
              f"{starts[min(255, nwg - 1)]:.1f} / {starts[min(511, nwg - 1)]:.1f} / {starts[min(767, nwg - 1)]:.1f} us)")


if __name__ == "__main__":
    main()
