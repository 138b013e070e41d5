#!/bin/bash
# Stream drivers side by side: one host thread with split calls vs one blocking host thread per stream.
out=gpurun_out/${1:-driver}; mkdir -p $out
python -m pytest tests/test_parity_gpu.py -q -m gpu -k "multi_stream or split_call" -x > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
for wl in c3 c2; do
  for cfg in "pipelined 2" "pipelined 3" "pipelined 4" "threads 3" "pipelined 3" "threads 3"; do
    set -- $cfg
    timeout 300 python bench.py --workload $wl --driver $1 --streams $2 --steps 240 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl $1 $2', round(d['value'],1), d['ms_per_step'])" >> $out/rates.txt
  done
done
cat $out/status.txt $out/rates.txt; tail -5 $out/pytest.log
