set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_fp8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fp8 -- python3 bench.py --model vit_l --batch 128 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
find gpurun_out/prof_fp8 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_vit_l_fp8_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r02_vit_l_fp8_kernel_stats.csv")))
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6/11:8.3f} ms/step  {int(r["Calls"])//11:4d}/step  {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:150]}')
PY
