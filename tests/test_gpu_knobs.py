"""Every environment knob of the HIP library that selects another kernel or schedule (README, "Developer A/B knobs") is reachable
by a user through the environment, so it gets the same op-parity cases as the default path: the SDPA / NT-GEMM / TN-GEMM tests of
tests/test_gpu_ops.py and tests/test_gpu_round3.py run again with the knob set.  One fresh child process per knob: the library reads
a knob once and caches it (csrc/common.h VsEnvInt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NT = "test_gemm_nt_epilogues or test_nt_line_shaped_epilogue or test_gemm_nt_embed_epilogue"
TN = "test_gemm_tn"
ATTN = "test_attention_fwd_bwd or test_attention_large_logits or test_attention_golden"

FP8_TN = "test_gemm_fp8_tn_weight_gradient or test_gemm_fp8_tn_batch_matches_single_launches"

KNOBS = [
    ({"VITSSL_NT_PP": "0"}, NT),                      # the two-phase NT loop instead of the ping-pong loop
    ({"VITSSL_NT_STAGGER": "0", "VITSSL_NT_PERSIST": "1"}, NT),   # no start-up stagger; persistent two-phase loop for every K
    ({"VITSSL_NT_TILE": "3", "VITSSL_NT_GROUPN": "2"}, NT),       # 192-row tiles everywhere, raster groups of two tile columns
    ({"VITSSL_TN_PP": "0"}, TN),                      # the two-phase weight-gradient loop
    ({"VITSSL_TN_BATCH_REM": "0", "VITSSL_TN_BATCH_SPLITS": "3"}, TN),   # batched weight gradients: no helper workgroups, forced split count
    ({"VITSSL_ATTN_FWD_PERSIST": "0", "VITSSL_ATTN_BWD_PIPE": "0", "VITSSL_ATTN_STAGGER_BWD": "0"}, ATTN),   # N > 128 on the short-sequence kernels
    ({"VITSSL_ATTN_BWD_PERSIST": "0"}, ATTN),         # 129-224 tokens: one workgroup per (batch, head) with the pipelined prologue
    ({"VITSSL_TN8_PP": "0"}, FP8_TN),                 # e4m3 weight gradients with all eight waves in step (the round-2 loop)
]


@pytest.mark.parametrize("env,select", KNOBS, ids=[" ".join(f"{k}={v}" for k, v in e.items()) for e, _ in KNOBS])
def test_op_parity_holds_under_knob(env, select):
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), os.path.join(ROOT, "tests", "test_gpu_round3.py"),
           os.path.join(ROOT, "tests", "test_gpu_fp8.py"), "-x", "-q", "-m", "gpu", "-k", select, "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env), timeout=900, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail
