#!/bin/bash
# Per-epilogue counter table of the NT GEMM at the ViT-B/16 SimMIM shapes (M = 50176), model-faithful
# arguments (dropout 0.1 on EPI 2/3, bias-gradient column sums on EPI 4).  Separate rocprofv3 --pmc passes
# (SQ set, FETCH_SIZE, WRITE_SIZE) plus one --kernel-trace pass for the duration.
#   usage (GPU box): bash tools/epilogue_pmc.sh r02   ->  gpurun_out/r02_epilogue_pmc.json
set -e
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/epi_pmc_$tag
rm -rf $out
cases="0:2304:768 0:768:3072 2:3072:768 3:768:768 3:768:3072 4:3072:768"
for c in $cases; do
  IFS=: read epi N K <<< "$c"
  d=$out/e${epi}_${N}_${K}
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/t -- python3 tools/one_gemm.py nt $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $d/a -- python3 tools/one_gemm.py nt $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/f -- python3 tools/one_gemm.py nt $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/w -- python3 tools/one_gemm.py nt $N $K $epi > /dev/null 2>&1
  echo "done $c" >&2
done
python3 - "$out" "gpurun_out/${tag}_epilogue_pmc.json" <<'PY'
import csv, glob, json, os, sys, collections
out, dst = sys.argv[1], sys.argv[2]
res = {}
M = 50176
for d in sorted(glob.glob(out + "/e*")):
    epi, N, K = (int(v) for v in os.path.basename(d)[1:].split("_"))
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/[afw]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(d + "/t/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    row = {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in agg.items()}      # skip the two warm-up launches
    dur = sorted(dur[2:])
    us = dur[len(dur) // 2] if dur else None
    elt_out = {0: 2, 2: 4, 3: 4, 4: 2}[epi]
    alg = M * K * 2 + N * K * 2 + M * N * elt_out + (M * N * 4 if epi == 3 else 0) + (M * N * 2 if epi == 4 else 0)
    hbm = (2 * row.get("FETCH_SIZE", 0) + row.get("WRITE_SIZE", 0)) * 1024
    res[f"epi{epi}_N{N}_K{K}"] = dict(us=us, tflops=None if not us else round(2.0 * M * N * K / us / 1e6, 1), alg_bytes=alg,
                                      hbm_bytes_pmc=round(hbm), traffic_ratio=round(hbm / alg, 3), counters=row,
                                      mfma_busy_frac=None if not us else round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (us * 2400.0), 3),
                                      wait_any_frac=None if "SQ_WAVE_CYCLES" not in row else round(row.get("SQ_WAIT_ANY", 0) / max(row["SQ_WAVE_CYCLES"], 1), 3))
    r = res[f"epi{epi}_N{N}_K{K}"]
    if us and row.get("SQ_BUSY_CYCLES"):     # the clock the chip held over the launch, and the matrix pipes' share of THOSE cycles
        r["clock_ghz"] = round(row["SQ_BUSY_CYCLES"] / 32 / us / 1e3, 3)
        r["mfma_busy_frac_at_clock"] = round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (us * 1e3 * r["clock_ghz"]), 3)
res["note"] = ("rocprofv3 passes of tools/one_gemm.py (6 launches, first two dropped); FETCH_SIZE doubled (gfx950 wide reads, "
               "MI355X_MICROARCH.md), unit KB; SQ_* summed over the chip; us = median kernel duration of the --kernel-trace pass; "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES (16 per 16x16x32 MFMA, summed over SIMDs) / 1024 SIMDs / (us x 2.4 GHz peak clock); "
               "clock_ghz = SQ_BUSY_CYCLES (one instance per shader engine, 32 on the chip) / 32 / us: within 3 % of the in-kernel "
               "s_memtime / s_memrealtime clock of tools/nt_stamps.py; mfma_busy_frac_at_clock uses that clock instead of 2.4 GHz")
json.dump(res, open(dst, "w"), indent=1)
for k, v in res.items():
    if k != "note":
        print(k, v["us"], v["tflops"], v["traffic_ratio"], v["mfma_busy_frac"], v["wait_any_frac"], v.get("clock_ghz"), v.get("mfma_busy_frac_at_clock"))
PY
