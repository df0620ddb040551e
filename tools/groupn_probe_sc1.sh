cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02_groupn_sc1.txt; : > $o
for lib in aux16; do for g in 4 6 12; do
  for c in "2 3072 768" "0 2304 768"; do
    set -- $c
    d=gpurun_out/groupn/${lib}_g${g}_e$1_$2
    rm -rf $d
    VITSSL_LIB=$PWD/tools/build/libvitssl_$lib.so VITSSL_NT_GROUPN=$g rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/one_gemm.py nt $2 $3 $1 > /dev/null 2>&1
    python3 - $d "$lib g$g epi$1 N$2 K$3" >> $o <<'PY'
import csv, glob, sys
v = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            v.append(float(r["Counter_Value"]))
v = v[2:]
print(f"{sys.argv[2]}: FETCH_SIZE x2 = {2 * sum(v) / max(len(v), 1) / 1024:.1f} MB per launch (n={len(v)})")
PY
  done
done; done
cat $o
