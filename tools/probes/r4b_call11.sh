mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py tests/test_gpu_fuzz.py -q 2>&1 | grep -v Warn | tail -15
out=gpurun_out/r4b_fuzz2.txt; : > $out
sweep() { echo "## $1" >> $out; shift; env "$@" python tools/fuzz_ops.py 2>&1 | grep -v Warn | tail -5 | cut -c1-700 >> $out; }
sweep "SEED=22 KINDS=simdrop BUDGET_S=180 (1-3 block SimMIM models, dropout 0.05-0.5 ON through the fused train_step, exported masks, same dpred, bar 2e-2)" SEED=22 KINDS=simdrop BUDGET_S=180
cat $out | cut -c1-330
