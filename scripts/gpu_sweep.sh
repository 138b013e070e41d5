#!/bin/bash
# sweep of bench options.  usage: scripts/gpu_sweep.sh tag  (edit the list below)
set -u
TAG=${1:-sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 3 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], d['roofline']['frame']['single_stream_ms_p50'], {k:v['ms'] for k,v in d['roofline']['stages'].items()}, d['roofline']['slab_pairs_last_frame'])"; }
run c2_auto --workload c2
run c2_slabs1 --workload c2 --slabs 1
run c2_slabs1_nodefer --workload c2 --slabs 1 --no-defer-colour
run c2_auto_s5 --workload c2 --streams 5
run c2_slabs1_nodefer_s5 --workload c2 --slabs 1 --no-defer-colour --streams 5
run c3_first300 --slab-first 300
run c3_first600 --slab-first 600
run c3_first800 --slab-first 800
run c3_s4 --streams 4
run c3_s5 --streams 5
run c3_s6 --streams 6
run heavy_auto --workload heavy
run heavy_slabs1 --workload heavy --slabs 1
run heavy_first200 --workload heavy --slab-first 200
run heavy_first800 --workload heavy --slab-first 800
run heavy1080_slabs1 --workload heavy1080 --slabs 1
