#!/bin/bash
# second counter set: texture-addresser / L2 busy, VMEM queue pressure
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/pmc2_$tag
rm -rf $out
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $out/a -- python tools/one_gemm.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/b -- python tools/one_gemm.py "$@" > /dev/null 2>&1
rocprofv3 --pmc GRBM_TC_BUSY GRBM_EA_BUSY --output-format csv -d $out/c -- python tools/one_gemm.py "$@" > /dev/null 2>&1
rocprofv3 --pmc TCC_BUSY_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $out/d -- python tools/one_gemm.py "$@" > /dev/null 2>&1 || true
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k][2:] if len(agg[k]) > 3 else agg[k]
    print(f"{k:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
