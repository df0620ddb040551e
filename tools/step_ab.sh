o=gpurun_out/r2_stepab.txt; : > $o
for r in 1 2; do
for cfg in "main 0" "bf16sc1 6" "bf16sc1 0" "main 6"; do
  set -- $cfg
  L=""; [ $1 != main ] && L=$PWD/tools/build/libvitssl_$1.so
  VITSSL_LIB=$L VITSSL_NT_GROUPN=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 g$2', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['families'].items()})" >> $o
done; done
cat $o
