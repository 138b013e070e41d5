#!/bin/bash
# Headline rate against the number of streams, for the driver's short region (K = 20) and a long one (K = 200), same box.
# usage: scripts/gpu_streams_sweep.sh  -> gpurun_out/streams_sweep.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for K in 20 200; do
  for S in 3 5 7 9 11 14; do
    timeout 300 python bench.py --steps $K --warmup 5 --streams $S --no-cpu-baseline --no-also --no-reference-hip 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K', $K, 'S', $S, d['value'], d['ms_per_step'], d['regions']['values'])"
  done
done | tee gpurun_out/streams_sweep.txt
