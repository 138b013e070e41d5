#!/bin/bash
# A/B two builds of the library on the same box: per-stage times, single stream.  usage: gpu_ab_lib.sh tag libA libB
out=gpurun_out/${1:-ablib}; mkdir -p $out
for rep in 1 2 3; do
for lib in "$2" "$3"; do
GSR_LIB=$PWD/$lib timeout 300 python bench.py --streams 1 --steps 120 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']; print('$lib', round(d['value'],1), ' '.join(k+'='+str(v['ms']) for k,v in st.items()))" >> $out/ab.txt
done; done
cat $out/ab.txt
