#!/bin/bash
# What the driver does at round end, in its order and with its command lines: pytest -m gpu -x, smoke(), bench.py --gpus 1 --steps 20 --warmup 5.
out=gpurun_out/${1:-r5g}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
( time python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" ) > $out/smoke.log 2>&1; echo "smoke exit $?" >> $out/status.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench exit $?" >> $out/status.txt
tail -4 $out/pytest.log; tail -6 $out/smoke.log; tail -4 $out/bench_steps20.err; cat $out/status.txt
python -c "
import json; d=json.loads(open('$out/bench_steps20.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')}); print(d['config'].get('workload')); print(d['roofline']['frac'], d['roofline']['traffic_source'], d['cpu_baseline']['value'], d['cpu_baseline'].get('torch_cpu_c3_seconds_per_frame'))"
