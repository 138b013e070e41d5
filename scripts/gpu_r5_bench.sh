#!/bin/bash
# Round 5: the default bench line on the final tree; grad slabs on a trained-scene-like cloud.
out=gpurun_out/${1:-r5j}; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for w in heavy heavy1080; do for s in 0 1; do
  GSR_GRAD_SLABS=$s timeout 200 python scripts/bench_backward.py --workload $w --steps 30 2>/dev/null | tail -1 | sed "s/^/slabs=$s /" >> $out/bw_heavy_ab.txt
done; done; done
cat $out/bw_heavy_ab.txt
( time timeout 900 python bench.py ) > $out/bench.json 2> $out/bench.err; echo "bench exit $?" > $out/status.txt
tail -4 $out/bench.err; cat $out/status.txt
