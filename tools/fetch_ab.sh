cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for sg in 0 1; do
  export VITSSL_NT_STAGGER=$sg
  rm -rf gpurun_out/fetch_sg$sg
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_sg$sg -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-configs > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
for sg in (0, 1):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/fetch_sg{sg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE" or "gemm_nt_pp" not in r["Kernel_Name"]:
                continue
            m = re.search(r"gemm_nt_pp_kernel<(\d+), .*NtCfg<64, 2, 4, (\d+)>", r["Kernel_Name"])
            k = m.groups() if m else r["Kernel_Name"][:40]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    print("stagger", sg, {k: (round(v[0] / v[1] / 1024 * 2), v[1]) for k, v in sorted(agg.items())}, "MB per launch (x2 corrected), launches")
PY
