#!/bin/bash
# Cold-process hunt of the round-2 failure: the first GPU test of the suite, N fresh processes per setting of the radix rank
# (LDS adds / ballots), failures kept with their dumps (gpurun_out/failures/).  usage: gpu_r3_hunt.sh tag runs
set -u
TAG=${1:-hunt}; RUNS=${2:-25}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rank in 1 0; do
  fails=0
  for i in $(seq 1 $RUNS); do
    GSR_RADIX_RANK=$rank timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "test_backward_sh_scene or test_backward_precomputed" > $OUT/run.log 2>&1
    if ! grep -q " passed" $OUT/run.log || grep -q "failed" $OUT/run.log; then fails=$((fails+1)); cp $OUT/run.log $OUT/fail_rank${rank}_$i.log; grep -E "^E  " $OUT/run.log | head -6; fi
  done
  echo "rank $rank (1 = LDS adds, 0 = ballots): cold runs $RUNS failures $fails" | tee -a $OUT/status.txt
done
ls gpurun_out/failures 2>/dev/null | head
