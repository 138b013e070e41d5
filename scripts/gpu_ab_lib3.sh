#!/bin/bash
# A/B two builds of the library on the same box, 3 streams (the bench default).  usage: gpu_ab_lib3.sh tag libA libB
out=gpurun_out/${1:-ablib3}; mkdir -p $out
for rep in 1 2 3 4; do
for lib in "$2" "$3"; do
GSR_LIB=$PWD/$lib timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), d['ms_per_step'])" >> $out/ab.txt
done; done
cat $out/ab.txt
