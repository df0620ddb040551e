set -e
export EPIS=bf16,resid+drop ROUNDS=9 ITERS=3
echo "== N=768: default at M=50176 (224-row, 2.625 rounds)"
M=50176 SHAPES="768,768;768,3072;768,2304" python tools/ab_gemm.py
echo "== N=768: 192-row at M=49152 (3.0 rounds exactly)"
VITSSL_NT_TILE=3 M=49152 SHAPES="768,768;768,3072;768,2304" python tools/ab_gemm.py
echo "== N=768: 192-row at M=48960 (765 tiles)"
VITSSL_NT_TILE=3 M=48960 SHAPES="768,768;768,3072;768,2304" python tools/ab_gemm.py
export EPIS=bf16,gelu+drop,dgelu+drop
echo "== N=3072: default at M=50176 (256-row 9.19 rounds)"
M=50176 SHAPES="3072,768" python tools/ab_gemm.py
echo "== N=3072: 224-row at M=47712 (2556 tiles, 9.98 rounds)"
VITSSL_NT_TILE=4 M=47712 SHAPES="3072,768" python tools/ab_gemm.py
echo "== N=3072: 256-row at M=54528 (213 panels 2556 tiles)"
VITSSL_NT_TILE=1 M=54528 SHAPES="3072,768" python tools/ab_gemm.py
