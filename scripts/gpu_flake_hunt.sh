#!/bin/bash
# Repeat one GPU test in a fresh process (whole-suite collection, as the round-end run has it) and keep the failures.
# usage: gpu_flake_hunt.sh tag "<-k expression>" runs [GSR_LIB path]
set -u
TAG=$1; K=$2; N=$3; LIB=${4:-}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$LIB" ] && export GSR_LIB=$PWD/$LIB
fails=0
for i in $(seq 1 $N); do
  timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "$K" > $OUT/run.log 2>&1
  if ! grep -q " passed" $OUT/run.log || grep -q "failed" $OUT/run.log; then
    fails=$((fails+1)); cp $OUT/run.log $OUT/fail_$i.log; grep -E "^E  " $OUT/run.log | head -8
  fi
done
echo "runs $N failures $fails (lib: ${LIB:-default})"
