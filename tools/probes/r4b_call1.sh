set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k layernorm 2>&1 | tail -3
echo "== LN A/B D=384"
LIBS=main,ln1row,main,ln1row D=384 python tools/ab_ln.py
echo "== LN A/B D=384 M=25088"
LIBS=main,ln1row D=384 M=25088 python tools/ab_ln.py
echo "== ViT-S bench pairs / one row"
python bench.py --model vit_s --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs | tail -1 | cut -c1-400
VITSSL_LIB=tools/build/libvitssl_ln1row.so python bench.py --model vit_s --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs | tail -1 | cut -c1-400
python bench.py --model vit_s --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs | tail -1 | cut -c1-400
echo "== fuzz sim8"
KINDS=sim8 BUDGET_S=90 SEED=7 python tools/fuzz_ops.py 2>&1 | tail -4
