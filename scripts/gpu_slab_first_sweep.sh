#!/bin/bash
# GSR_OPT_SLAB_FIRST (pairs per tile, on average, in the first depth slab) against the headline and the serial rate, same box.
# usage: scripts/gpu_slab_first_sweep.sh [values...]  -> gpurun_out/slab_first_sweep.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
VALS=${@:-"250 325 400 500 650"}
for rep in 1 2; do
  for v in $VALS; do
    timeout 300 python bench.py --steps 40 --warmup 10 --slab-first $v --no-cpu-baseline --no-also --no-reference-hip 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slab_first', $v, 'rep', $rep, d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], d['roofline']['slab_pairs_last_frame'])"
  done
done | tee gpurun_out/slab_first_sweep.txt
