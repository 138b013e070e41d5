#!/bin/bash
# parity tests + render()-boundary bench lines (single stream and 3 streams)
out=gpurun_out/${1:-quick4}; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
for s in 1 3; do
timeout 300 python bench.py --boundary render --streams $s --steps 240 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('render boundary streams $s', round(d['value'],1), d['ms_per_step'])" >> $out/rates.txt
done
cat $out/status.txt $out/rates.txt; tail -4 $out/pytest.log
