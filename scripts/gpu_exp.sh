#!/bin/bash
set -u
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 5 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], d['ms_per_step'], d['roofline']['frame']['single_stream_ms_p50'])"; }
for rep in 1 2; do
run base
GSR_EXPERIMENT_EMPTY_LAUNCHES=10 run empty10
GSR_EXPERIMENT_EMPTY_LAUNCHES=20 run empty20
done
