#!/bin/bash
# Round 3 soak: the whole GPU suite N times in fresh processes (failures kept), then the multi-stream race hunt.
# usage: gpu_r3_soak.sh tag [runs]
set -u
TAG=${1:-soak}; RUNS=${2:-4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
fails=0
for i in $(seq 1 $RUNS); do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/run_$i.log 2>&1
  tail -1 $OUT/run_$i.log
  if grep -q "failed" $OUT/run_$i.log || ! grep -q " passed" $OUT/run_$i.log; then fails=$((fails+1)); else rm -f $OUT/run_$i.log; fi
done
echo "full-suite runs $RUNS, runs with failures $fails" | tee $OUT/status.txt
timeout 900 python scripts/stress_streams.py > $OUT/stress_streams.txt 2>&1; tail -6 $OUT/stress_streams.txt
