cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_dino
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dino -- python3 tools/bench_dino.py > gpurun_out/dino_prof.log 2>&1
find gpurun_out/prof_dino -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/dino_stats.csv
tail -1 gpurun_out/dino_prof.log
