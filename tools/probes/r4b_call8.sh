mkdir -p gpurun_out
echo "== 4-token models"
python tools/probes/repro_simmim.py "3 16 8 128 2 128" "1 32 16 64 1 64" "5 16 8 192 3 256" "2 32 16 384 6 128" "4 16 8 64 1 320" "6 32 16 128 2 64" 2>&1 | grep -v Warn
KINDS=sim8 BUDGET_S=100 SEED=13 python tools/fuzz_ops.py > gpurun_out/fz_sim8.log 2>&1; grep -A8 FAILED gpurun_out/fz_sim8.log | tail -12; tail -3 gpurun_out/fz_sim8.log | cut -c1-200
KINDS=simmim,simdrop BUDGET_S=100 SEED=14 python tools/fuzz_ops.py > gpurun_out/fz_s14.log 2>&1; grep -A8 FAILED gpurun_out/fz_s14.log | tail -12; tail -3 gpurun_out/fz_s14.log | cut -c1-250
