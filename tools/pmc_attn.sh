#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> <kind> <N> <K> [epi]   -- SQ + TCC counters of one GEMM shape
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmc_$tag
rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/a -- python tools/one_attn.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $out/b -- python tools/one_attn.py "$@" > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/c -- python tools/one_attn.py "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/d -- python tools/one_attn.py "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/e -- python tools/one_attn.py "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections, os
agg = collections.defaultdict(list)
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KSEL","attn_fwd") in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k][2:] if len(agg[k]) > 3 else agg[k]
    print(f"{k:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
