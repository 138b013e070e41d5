#!/bin/bash
# Round 5: the driver's test command, the deterministic backward's cost with a kernel trace, the truth report, one bench line.
out=gpurun_out/${1:-r5b}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
tail -15 $out/pytest.log
for w in c2 c3; do
  timeout 200 python scripts/bench_backward.py --workload $w --steps 20 > $out/bw_$w.json 2> $out/bw_$w.err
  GSR_BACKWARD_DETERMINISTIC=1 timeout 200 python scripts/bench_backward.py --workload $w --steps 20 > $out/bw_${w}_det.json 2> $out/bw_${w}_det.err
done
( cd /tmp && GSR_BACKWARD_DETERMINISTIC=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_det -o det -- python $GRAFT_REPO_ROOT/scripts/bench_backward.py --workload c3 --steps 10 > $GRAFT_REPO_ROOT/$out/prof_det.log 2>&1 )
find $out/prof_det -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $out/det_c3_kernel_stats.csv
rm -rf $out/prof_det
timeout 900 python scripts/gradient_truth_report.py > $out/gradient_truth.md 2> $out/gradient_truth.err; echo "truth exit $?" >> $out/status.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/status.txt
cat $out/status.txt; cat $out/bw_*.json; head -12 $out/det_c3_kernel_stats.csv
