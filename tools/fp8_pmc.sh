#!/bin/bash
# Counter table of the e4m3-operand GEMMs at the ViT-L/16 batch-128 shapes (M = 25088): separate rocprofv3 --pmc passes
# (SQ set, FETCH_SIZE, WRITE_SIZE) plus one --kernel-trace pass for the duration.
#   usage (GPU box): bash tools/fp8_pmc.sh r02   ->  gpurun_out/r02_fp8_pmc.json
set -e
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fp8_pmc_$tag
rm -rf $out
cases="nt:3072:1024:0 nt:1024:4096:0 nt:4096:1024:2 nt:1024:4096:3 nt:4096:1024:4 tn:4096:1024:0 tn:1024:1024:0"
for c in $cases; do
  IFS=: read kind N K epi <<< "$c"
  d=$out/${kind}_${N}_${K}_${epi}
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/t -- python3 tools/one_fp8_gemm.py $kind $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $d/a -- python3 tools/one_fp8_gemm.py $kind $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/f -- python3 tools/one_fp8_gemm.py $kind $N $K $epi > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/w -- python3 tools/one_fp8_gemm.py $kind $N $K $epi > /dev/null 2>&1
  echo "done $c" >&2
done
python3 - "$out" "gpurun_out/${tag}_fp8_pmc.json" <<'PY'
import csv, glob, json, os, sys, collections
out, dst = sys.argv[1], sys.argv[2]
res = {}
M = 25088
for d in sorted(glob.glob(out + "/*_*")):
    kind, N, K, epi = os.path.basename(d).split("_")
    N, K, epi = int(N), int(K), int(epi)
    want = "gemm_nt_pp" if kind == "nt" else "gemm_tn_fp8"
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/[afw]/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(d + "/t/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    row = {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in agg.items()}      # skip the two warm-up launches
    dur = sorted(dur[2:])
    us = dur[len(dur) // 2] if dur else None
    if kind == "nt":
        out_b = {0: 2, 2: 2 + 1, 3: 4, 4: 1}[epi]                 # bytes written per output element
        in_b = {0: 0, 2: 0, 3: 4, 4: 2}[epi]                      # epilogue bytes read per output element
        alg = M * K + N * K + M * N * (out_b + in_b)
        flops = 2.0 * M * N * K
    else:
        alg = M * N + M * K + N * K * 4                             # both operand images + the fp32 gradient (slabs not counted)
        flops = 2.0 * M * N * K
    hbm = (2 * row.get("FETCH_SIZE", 0) + row.get("WRITE_SIZE", 0)) * 1024
    # MFMA busy: the counter adds 32 cycles per K = 128 MFMA and SIMD (measured: cycles = #MFMA x 32)
    res[os.path.basename(d)] = dict(us=us, tflops=None if not us else round(flops / us / 1e6, 1), frac_of_5pf=None if not us else round(flops / us / 1e6 / 5000.0, 3),
                                    alg_bytes=alg, hbm_bytes_pmc=round(hbm), traffic_ratio=round(hbm / alg, 3), counters=row,
                                    mfma_busy_frac=None if not us else round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (us * 2400.0), 3))
res["note"] = ("rocprofv3 passes of tools/one_fp8_gemm.py (6 launches, first two dropped); FETCH_SIZE doubled (gfx950 wide reads, "
               "MI355X_MICROARCH.md), unit KB; SQ_* summed over the chip; us = median kernel duration of the --kernel-trace pass; "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (us x 2.4 GHz)")
json.dump(res, open(dst, "w"), indent=1)
for k, v in res.items():
    if k != "note":
        print(k, v["us"], v["tflops"], v["frac_of_5pf"], v["traffic_ratio"], v["mfma_busy_frac"])
PY
