#!/bin/bash
# parity tests + C3 bench lines + backward iteration time
out=gpurun_out/${1:-quick3}; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
for i in 1 2; do
timeout 300 python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); st=d['roofline']['stages']; print('c3', round(d['value'],1), d['ms_per_step'], 'pre', st['preprocess']['ms'], 'blend', st['blend']['ms'], 'p50', d['roofline']['frame']['single_stream_ms_p50'])" >> $out/rates.txt
done
timeout 300 python scripts/bench_backward.py --workload c3 > $out/backward.txt 2>&1
cat $out/status.txt $out/rates.txt; tail -3 $out/pytest.log; tail -6 $out/backward.txt
