mkdir -p gpurun_out
python tools/probes/repro_simdrop.py 5 48 16 384 6 384 3 0.05 2>&1 | grep -v Warning | tail -60
echo "== other candidates (assert mode)"
for c in "5 48 16 128 2 384 3 0.1" "3 64 16 128 2 64 3 0.1" "5 88 8 192 3 128 3 0.05"; do FUZZ_VERBOSE= python - $c <<'P' 2>&1 | tail -2
import os, sys
sys.path.insert(0, "tools")
os.environ.pop("FUZZ_VERBOSE", None)
import fuzz_ops
a = sys.argv[1:]
args = (int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), float(a[7]))
try:
    fuzz_ops.simmim_drop_case(fuzz_ops.ops, *args); print("ok", args)
except AssertionError as e:
    print("FAILED", args, e)
P
done
echo "== fuzz sim8"
KINDS=sim8 BUDGET_S=100 SEED=11 python tools/fuzz_ops.py > gpurun_out/fz_sim8.log 2>&1; grep -A8 FAILED gpurun_out/fz_sim8.log | tail -12; tail -1 gpurun_out/fz_sim8.log | cut -c1-200
