#!/bin/bash
# Round 5, first GPU call: the driver's own test command, the gradient truth report, the cost of the deterministic backward.
out=gpurun_out/${1:-r5a}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
tail -25 $out/pytest.log
timeout 900 python scripts/gradient_truth_report.py > $out/gradient_truth.md 2> $out/gradient_truth.err; echo "truth exit $?" >> $out/status.txt
for w in c2 c3; do
  timeout 200 python scripts/bench_backward.py --workload $w --steps 20 > $out/bw_$w.json 2> $out/bw_$w.err
  GSR_BACKWARD_DETERMINISTIC=1 timeout 200 python scripts/bench_backward.py --workload $w --steps 20 > $out/bw_${w}_det.json 2> $out/bw_${w}_det.err
done
cat $out/status.txt; tail -2 $out/bw_*.json; tail -3 $out/gradient_truth.err
