mkdir -p gpurun_out
python tools/probes/repro_simdrop.py 1 24 8 128 2 320 3 0.5 2>&1 | grep 'simdrop\|w_query\|w_key\|w_value\|final_linear\|linear_in.weight\|Error\|error'
KINDS=simmim,simdrop BUDGET_S=240 SEED=12 python tools/fuzz_ops.py > gpurun_out/fz_s12.log 2>&1
grep -B2 -A12 FAILED gpurun_out/fz_s12.log | grep -v Warning | tail -30
tail -1 gpurun_out/fz_s12.log | cut -c1-300
KINDS=sim8 BUDGET_S=100 SEED=13 python tools/fuzz_ops.py > gpurun_out/fz_sim8.log 2>&1; grep -A8 FAILED gpurun_out/fz_sim8.log | tail -12; tail -1 gpurun_out/fz_sim8.log | cut -c1-200
