#!/bin/bash
# In-step A/B of library variants: alternating bench.py runs (no CPU leg).  usage: bash tools/step_ab.sh "main gp16 gp2" [rounds]
o=gpurun_out/r2_stepab.txt; : > $o
for r in $(seq 1 ${2:-2}); do
for v in $1; do
  L=""; [ $v != main ] && L=$PWD/tools/build/libvitssl_$v.so
  VITSSL_LIB=$L timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['families'].items()})" >> $o
done; done
cat $o
