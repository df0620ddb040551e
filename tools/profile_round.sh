#!/bin/bash
# Regenerates the evidence kept under profiles/ for one round (run on the GPU box):
#   gpurun_out/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py` (11 steps)
#   gpurun_out/<tag>_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE / SQ_BUSY_CYCLES (+ dispatch durations: the clock held) per kernel
#                                       family, collected in SEPARATE --pmc passes of the same command (MI355X_MICROARCH.md)
#   gpurun_out/<tag>_bench.json         the plain bench line (with cpu_baseline), taken last so that it carries the PMC traffic
# usage: bash tools/profile_round.sh r01_final
set -e
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-other-configs > /dev/null 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_BUSY_CYCLES; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-other-configs > /dev/null 2>&1
done
python3 - "$out" "gpurun_out/${tag}_pmc_traffic.json" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
fam = lambda n: ("gemm_nt" if "gemm_nt" in n else "gemm_tn" if "gemm_tn" in n else "attn" if "attn_" in n
                 else "ln" if "ln_" in n else "other")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: {"sum_kb": 0.0, "dispatches": 0})
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            a = agg[fam(r["Kernel_Name"])]
            a["sum_kb"] += float(r["Counter_Value"])
            a["dispatches"] += 1
    res[c] = agg
# the clock held per kernel family: SQ_BUSY_CYCLES (one instance per shader engine, 32 on the chip) against the durations of
# the same dispatches in the same pass (bench.pmc_clock_ghz divides)
agg = collections.defaultdict(lambda: {"sum": 0.0, "duration_ns": 0.0, "dispatches": 0})
for f in glob.glob(f"{out}/SQ_BUSY_CYCLES/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_BUSY_CYCLES" or "Start_Timestamp" not in r:
            continue
        a = agg[fam(r["Kernel_Name"])]
        a["sum"] += float(r["Counter_Value"])
        a["duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a["dispatches"] += 1
res["SQ_BUSY_CYCLES"] = agg
sys.path.insert(0, ".")
import bench
res["kernel_sources_sha16"] = bench.kernel_sources_sha16()     # bench.py only quotes a summary of THESE kernel sources
res["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1`; "
               "unit KB as reported; FETCH_SIZE must be doubled on gfx950 for wide coalesced reads (MI355X_MICROARCH.md)")
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps({c: {k: round(v["sum_kb"] / max(v["dispatches"], 1)) for k, v in res[c].items()} for c in ("FETCH_SIZE", "WRITE_SIZE")}))
print("clock held (GHz):", json.dumps({k: round(v["sum"] / 32.0 / v["duration_ns"], 3) for k, v in res["SQ_BUSY_CYCLES"].items() if v["duration_ns"] > 0}))
PY
# the plain bench line LAST, on the same box: bench.py quotes roofline.traffic only from a profiles/*_pmc_traffic.json stamped with
# these kernel sources, so the summary just made goes where it looks (the copy under profiles/ of the GPU box is scratch: commit
# the one merged back into gpurun_out/)
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
