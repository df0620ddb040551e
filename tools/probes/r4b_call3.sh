set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== fuzz sim8"
KINDS=sim8 BUDGET_S=120 SEED=11 python tools/fuzz_ops.py 2>&1 | tail -3
echo "== fuzz ln"
KINDS=ln BUDGET_S=40 SEED=5 python tools/fuzz_ops.py 2>&1 | tail -2
echo "== fuzz simmim,simdrop"
KINDS=simmim,simdrop BUDGET_S=150 SEED=12 python tools/fuzz_ops.py 2>&1 | tail -2
echo "== dino bench"
python tools/bench_dino.py | tail -1 | cut -c1-300
