mkdir -p gpurun_out
KINDS=simmim,simdrop BUDGET_S=200 SEED=12 python tools/fuzz_ops.py > gpurun_out/fz_s12.log 2>&1
grep -B2 -A12 FAILED gpurun_out/fz_s12.log | grep -v Warning | tail -30
tail -1 gpurun_out/fz_s12.log | cut -c1-300
line=$(grep -m1 'FAILED simdrop' gpurun_out/fz_s12.log | sed 's/FAILED simdrop(//; s/)//; s/,/ /g')
if [ -n "$line" ]; then echo "== isolated, verbose: $line"; python tools/probes/repro_simdrop.py $line 2>&1 | grep 'simdrop\|w_query\|w_key\|w_value\|Error\|error' ; fi
