#!/bin/bash
# Raster group width (tile columns whose weight panels are walked together) vs L2-miss traffic and time.
#   usage (GPU box): bash tools/groupn_probe.sh  ->  gpurun_out/r02_groupn.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02_groupn.txt; : > $o
for g in 3 4 6 12; do
  echo "== group_n $g (time)" >> $o
  VITSSL_NT_GROUPN=$g VITSSL_NT_STAGGER=0 SHAPES="3072,768;2304,768" timeout -k 10 120 python3 tools/bench_gemm.py nt 2>&1 | grep -E " bf16 | gelu\+drop | dgelu\+drop " >> $o
  for c in "2 3072 768" "0 2304 768"; do
    set -- $c
    d=gpurun_out/groupn/g${g}_e$1_$2
    rm -rf $d
    VITSSL_NT_GROUPN=$g rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/one_gemm.py nt $2 $3 $1 > /dev/null 2>&1
    python3 - $d "$g epi$1 N$2 K$3" >> $o <<'PY'
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v.append(float(r["Counter_Value"]))
v = v[2:]
print(f"group_n {sys.argv[2]}: FETCH_SIZE x2 = {2 * sum(v) / max(len(v), 1) / 1024:.1f} MB per launch (n={len(v)})")
PY
  done
done
cat $o
