#!/bin/bash
# parity tests (optionally a -k subset) + C3 / C2 bench lines.  usage: scripts/gpu_quick5.sh tag [pytest -k expr]
set -u
TAG=${1:-q}; K=${2:-}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$K" 2>&1 | tail -8 > $OUT/pytest_gpu.log
else timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 > $OUT/pytest_gpu.log; fi
tail -3 $OUT/pytest_gpu.log
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 5 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], d['regions']['value_min'], d['regions']['value_max'], d['roofline']['frame']['single_stream_ms_p50'], {k:v['ms'] for k,v in d['roofline']['stages'].items()})"; }
run c3
run c3_s1 --streams 1
run c2 --workload c2
run heavy --workload heavy
