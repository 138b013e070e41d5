#!/bin/bash
# Round 6: the driver's three steps as the driver runs them, then the default bench line for profiles/.
out=gpurun_out/${1:-r6final}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
( time python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" ) > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/status.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench20 exit $?" >> $out/status.txt
( time python bench.py ) > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/status.txt
tail -4 $out/pytest.log; tail -3 $out/smoke.log; cat $out/status.txt; tail -4 $out/bench_steps20.err
