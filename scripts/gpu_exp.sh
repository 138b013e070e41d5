#!/bin/bash
# experiment: hardware queue count x stream count (GPU_MAX_HW_QUEUES, default 4), same box, alternating
set -u
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { q=$1; s=$2
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 5 --streams $s 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=$q streams=$s', d['value'])"; }
for rep in 1 2; do
run 4 3; run 4 7; run 4 11; run 8 3; run 8 7; run 8 8; run 8 11; run 16 11; run 16 15; run 2 3; run 1 3
done
