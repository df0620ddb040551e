#!/bin/bash
# rocprofv3 --kernel-trace --stats of the configurations other than the headline one (run on the GPU box):
#   gpurun_out/<tag>_vit_s_kernel_stats.csv, _vit_l_kernel_stats.csv, _vit_l_fp8_kernel_stats.csv, _dino_kernel_stats.csv
# plus the plain bench lines in gpurun_out/<tag>_other_models.txt.   usage: bash tools/profile_models.sh r02
set -e
tag=${1:-rNN}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_models_$tag
rm -rf $out
log=gpurun_out/${tag}_other_models.txt
: > $log
run() {  # name, then the program and its arguments
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 "$@" > /dev/null 2>&1
  find $out/$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_${name}_kernel_stats.csv
  echo "profiled $name" >&2
}
python3 bench.py --model vit_s --no-cpu-baseline --no-other-configs >> $log 2>/dev/null
python3 bench.py --model vit_l --batch 128 --no-cpu-baseline --no-other-configs >> $log 2>/dev/null
python3 bench.py --model vit_l --batch 128 --dtype fp8 --no-cpu-baseline --no-other-configs >> $log 2>/dev/null
python3 tools/bench_dino.py >> $log 2>/dev/null
python3 tools/bench_dino.py --raw >> $log 2>/dev/null || true
run vit_s bench.py --model vit_s --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-timing
run vit_l bench.py --model vit_l --batch 128 --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-timing
run vit_l_fp8 bench.py --model vit_l --batch 128 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-timing
run dino tools/bench_dino.py --steps 6 --warmup 2
# the same lines as one JSON document (copy to profiles/<tag>_configs.json)
python3 - "$log" "gpurun_out/${tag}_configs.json" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
json.dump({"box": "one MI355X, one gpurun call", "configs": rows}, open(sys.argv[2], "w"), indent=1)
PY
cat $log
