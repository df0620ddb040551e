#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in the library (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: python tools/resource_usage.py > profiles/rNN_kernel_resource_usage.txt   (runs on the CPU box: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
        except OSError:
            continue
        if r.returncode == 0:
            return r.stdout.splitlines()
    return names


def main():
    print("kernel-resource-usage of every kernel (hipcc " + " ".join(ge.FLAGS) + " -Rpass-analysis=kernel-resource-usage)")
    print(f"{'file':14s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>14s} {'VGPR spill':>10s} {'SGPR spill':>10s} {'occ':>4s}  kernel")
    worst = 0
    for src in ge.SOURCES:
        if not src.endswith(".hip"):
            continue
        cmd = [ge.HIPCC] + ge.FLAGS + ge.PER_FILE_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c",
                                                                         os.path.join(ge.CSRC, src), "-o", "/dev/null"]
        t = subprocess.run(cmd, capture_output=True, text=True).stderr
        blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
        rows = []
        for b in blocks:
            name = b.split("\n")[0].split(" [-Rpass")[0].strip()
            g = lambda k: int(re.search(k + r": (\d+)", b).group(1))  # noqa: E731
            rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"),
                         g(r"Occupancy \[waves/SIMD\]")))
        for (name, *vals), dn in zip(rows, demangle([r[0] for r in rows])):
            dn = re.sub(r"\(anonymous namespace\)::", "", dn)
            dn = re.sub(r"\(.*$", "", dn) if len(dn) > 110 else dn
            print(f"{src:14s} {vals[0]:5d} {vals[1]:5d} {vals[2]:5d} {vals[3]:14d} {vals[4]:10d} {vals[5]:10d} {vals[6]:4d}  {dn[:120]}")
            worst = max(worst, vals[3])
    print(f"largest scratch size: {worst} bytes/lane")


if __name__ == "__main__":
    main()
