#!/bin/bash
out=gpurun_out/${1:-pad}; mkdir -p $out
for rep in 1 2; do
for pad in 0 2048 3600 5000; do
timeout 300 python bench.py --steps 300 --warmup 20 --blend-lds-pad $pad --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', round(d['value'],1), d['ms_per_step'], 'blend', d['roofline']['stages']['blend']['ms'])" >> $out/pad.txt
done; done
cat $out/pad.txt
