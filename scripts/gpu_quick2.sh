#!/bin/bash
# parity tests + one C3 bench line (no CPU baseline, no reference kernels)
out=gpurun_out/${1:-quick2}; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
for i in 1 2; do
timeout 300 python bench.py --steps 240 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', round(d['value'],1), d['ms_per_step'], 'blend', d['roofline']['stages']['blend']['ms'], 'p50', d['roofline']['frame']['single_stream_ms_p50'])" >> $out/rates.txt
done
cat $out/status.txt $out/rates.txt; tail -4 $out/pytest.log
