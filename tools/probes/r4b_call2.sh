set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
echo "== adamw / ema probe"
python tools/probes/adamw_probe.py
echo "== fuzz sim8"
KINDS=sim8 BUDGET_S=150 SEED=11 python tools/fuzz_ops.py 2>&1 | tail -6
echo "== fuzz ln"
KINDS=ln BUDGET_S=40 SEED=5 python tools/fuzz_ops.py 2>&1 | tail -3
