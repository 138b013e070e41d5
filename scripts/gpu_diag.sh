#!/bin/bash
# A/B of the inference-call options on the bench workload + parity tests.  usage: scripts/gpu_diag.sh tag [workload]
set -u
TAG=${1:-diag}; WL=${2:-c3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/diag_expand.py --workload $WL > $OUT/diag_$WL.txt 2>&1; echo "diag exit $?"
grep -v amdgpu.ids $OUT/diag_$WL.txt
for st in 3 1; do for s in 0 1; do for dc in 1 0; do
  extra=""; [ $dc = 0 ] && extra="--no-defer-colour"
  f=$OUT/bench_${WL}_st${st}_slabs${s}_defer${dc}
  timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-reference-hip --streams $st --slabs $s $extra > $f.json 2> $f.err || echo "bench failed $f"
  python -c "import json; d=json.load(open('$f.json')); print('streams=$st slabs=$s defer=$dc', d['value'], d['roofline']['frame']['single_stream_ms_p50'], {k:v['ms'] for k,v in d['roofline']['stages'].items()})"
done; done; done
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
