#!/bin/bash
# experiment: finish frames as their counters arrive (gsr_forward_ready) vs in lock step, same box, alternating
set -u
TAG=${1:-exp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 5 "$@" 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['regions']['values'])"; }
for rep in 1 2 3; do
run poll_k200
GSR_EXPERIMENT_NO_READY=1 run lockstep_k200
run poll_k20 --steps 20 --warmup 5
GSR_EXPERIMENT_NO_READY=1 run lockstep_k20 --steps 20 --warmup 5
run poll_k20_s3 --steps 20 --warmup 5 --streams 3
run poll_c2 --workload c2
GSR_EXPERIMENT_NO_READY=1 run lockstep_c2 --workload c2
done
