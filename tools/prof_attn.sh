cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_attn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_attn -- python3 tools/bench_gemm.py attn > /dev/null 2>&1
find gpurun_out/prof_attn -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/attn_stats.csv
grep attn gpurun_out/attn_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
