#!/bin/bash
# Round 5: grad-mode forward through the slab pipeline (GSR_OPT_GRAD_SLABS): the suite, then the training iteration with it on and off.
out=gpurun_out/${1:-r5h}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
tail -25 $out/pytest.log
for rep in 1 2; do for w in c2 c3; do for s in 0 1; do
  GSR_GRAD_SLABS=$s timeout 200 python scripts/bench_backward.py --workload $w --steps 30 2>/dev/null | tail -1 | sed "s/^/slabs=$s /" >> $out/bw_ab.txt
done; done; done
cat $out/status.txt $out/bw_ab.txt
