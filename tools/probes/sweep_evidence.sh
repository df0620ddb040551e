#!/bin/bash
# the sweeps recorded in profiles/rNN_fuzz_ops.txt (run on the GPU box): bash tools/probes/sweep_evidence.sh > gpurun_out/fuzz.txt
sweep() { echo "## $1"; shift; env "$@" python tools/fuzz_ops.py 2>&1 | grep -v Warn | tail -4 | cut -c1-700; }
sweep "SEED=31 BUDGET_S=120 (NT bf16 epilogues, TN, attention 1..256 tokens, LayerNorm, e4m3 NT / TN, batched TN)" SEED=31 BUDGET_S=120
sweep "SEED=32 KINDS=vit BUDGET_S=80 (whole supervised ViTs)" SEED=32 KINDS=vit BUDGET_S=80
sweep "SEED=33 KINDS=dino BUDGET_S=150 (whole DINO models; the teacher's store keeps no transposed weight images)" SEED=33 KINDS=dino BUDGET_S=150
