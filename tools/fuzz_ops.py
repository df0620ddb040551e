#!/usr/bin/env python3
"""Random-shape sweep of the op-parity checks (developer tool, GPU box): runs the bodies of tests/test_gpu_ops.py on
shapes the parametrised lists do not name -- ragged M / N, every N % 8 / N % 4 residue class the entry points accept,
token counts 1..256 -- and prints the first failing shape.

    SEED=1 BUDGET_S=150 python tools/fuzz_ops.py            (exit code 1 on the first failure)"""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_ops as T  # noqa: E402
import test_gpu_fp8 as T8  # noqa: E402
import test_gpu_round3 as T3  # noqa: E402
from vitssl_hip import ops  # noqa: E402


def main():
    rng = random.Random(int(os.environ.get("SEED", "0")))
    budget = float(os.environ.get("BUDGET_S", "120"))
    t0 = time.time()
    n = 0
    cases = []
    while time.time() - t0 < budget:
        kind = rng.choice((os.environ.get("KINDS") or "nt,nt,tn,attn,ln,nt8,tn8,tnb").split(","))
        if kind == "nt":
            big = rng.random() < 0.15
            M = rng.randint(1, 40000) if big else rng.randint(1, 3000)
            N = 4 * rng.randint(1, 200 if big else 700)
            K = 64 * rng.randint(1, 3 if big else 16)
            args = (M, N, K)
            fn = T.test_gemm_nt_epilogues
        elif kind == "tn":
            M = rng.randint(1, 9000)
            N1 = 8 * rng.randint(1, 150)
            N2 = 8 * rng.randint(1, 150)
            args = (M, N1, N2)
            fn = T.test_gemm_tn
        elif kind == "nt8":                                  # e4m3 operands: N % 8 == 0, K % 128 == 0
            args = (rng.randint(1, 3000), 8 * rng.randint(1, 300), 128 * rng.randint(1, 8))
            fn = lambda _ops, *a: T8.test_gemm_fp8_epilogues(*a)  # noqa: E731
        elif kind == "tn8":                                  # e4m3 weight gradients: N1, N2 % 16 == 0
            args = (rng.randint(1, 9000), 16 * rng.randint(1, 70), 16 * rng.randint(1, 70))
            fn = lambda _ops, *a: T8.test_gemm_fp8_tn_weight_gradient(*a)  # noqa: E731
        elif kind == "tnb":                                  # several weight gradients in one launch
            args = (rng.randint(1, 6000), [(8 * rng.randint(1, 200), 8 * rng.randint(1, 200)) for _ in range(rng.randint(1, 5))])
            fn = lambda _ops, *a: T3.test_gemm_tn_batch_matches_single_launches(*a)  # noqa: E731
        elif kind == "attn":
            args = (rng.randint(1, 256),)
            fn = T.test_attention_fwd_bwd
        else:
            args = (4 * rng.randint(1, 512),)
            fn = T.test_layernorm_fwd_bwd
        cases.append((kind, args))
        try:
            fn(ops, *args)
        except Exception:
            print(f"FAILED {kind}{args}", flush=True)
            traceback.print_exc()
            return 1
        n += 1
        if n % 10 == 0:
            print(f"{n} cases ok ({time.time() - t0:.0f} s); last: {kind}{args}", flush=True)
    print(f"all {n} cases ok: " + " ".join(f"{k}{a}" for k, a in cases[-12:]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
