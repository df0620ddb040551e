"""A short, deterministic slice of tools/fuzz_ops.py in the GPU suite: the op-parity and whole-model checks on shapes and
configurations drawn from a fixed seed (the long sweeps of round 4 are recorded in profiles/r04_fuzz_ops.txt).  Third slice:
SimMIM models with dropout ON through the fused train_step (exported masks) and on e4m3 operands."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("kinds,cases", [("nt,tn,attn,ln,nt8,tn8,tnb", 60), ("simmim,vit,dino", 40), ("simdrop,sim8", 24)])
def test_random_shapes_and_models_against_the_oracle(kinds, cases):
    import fuzz_ops
    assert fuzz_ops.run(seed=2026, kinds=kinds, budget_s=300.0, max_cases=cases) == 0
