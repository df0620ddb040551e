cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_now
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_now -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-other-configs > /dev/null 2>&1
find gpurun_out/prof_now -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_now.csv
