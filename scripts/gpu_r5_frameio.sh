#!/bin/bash
out=gpurun_out/${1:-r5m}; mkdir -p $out; export TMPDIR=/tmp
python scripts/bench_frame_files.py 960x540 > $out/frameio.jsonl 2> $out/frameio.err
python scripts/bench_frame_files.py 1920x1080 >> $out/frameio.jsonl 2>> $out/frameio.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o f -- python $GRAFT_REPO_ROOT/scripts/bench_frame_files.py 960x540 > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err )
F=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $out/frameio_kernel_stats.csv; rm -rf $out/prof
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
cat $out/frameio.jsonl; tail -4 $out/pytest.log; cat $out/status.txt; head -12 $out/frameio_kernel_stats.csv | cut -c1-150
