mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
out=gpurun_out/r4b_fuzz.txt; : > $out
sweep() { echo "## $1" >> $out; shift; env "$@" python tools/fuzz_ops.py 2>&1 | grep -v Warn | tail -5 | cut -c1-700 >> $out; }
sweep "SEED=21 KINDS=simmim BUDGET_S=150 (whole 2-block SimMIM models, 4..256 tokens, widths 64..384, judged for the engine's own dpred against the fp32 / bf16 / flash-backward modes of the oracle, bar 2e-2)" SEED=21 KINDS=simmim BUDGET_S=150
sweep "SEED=22 KINDS=simdrop BUDGET_S=180 (1-3 block SimMIM models, dropout 0.05-0.5 ON through the fused train_step, exported masks, same dpred, bar 2e-2)" SEED=22 KINDS=simdrop BUDGET_S=180
sweep "SEED=23 KINDS=sim8 BUDGET_S=120 (2-block SimMIM models on e4m3 operands against the oracle's fp8 mode with the engine's gradient scales)" SEED=23 KINDS=sim8 BUDGET_S=120
sweep "SEED=24 KINDS=ln BUDGET_S=40 (LayerNorm forward / backward, a quarter of the cases on the row-pair backward: 384 columns, even row count)" SEED=24 KINDS=ln BUDGET_S=40
cat $out | cut -c1-330
