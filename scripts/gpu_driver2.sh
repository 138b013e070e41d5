#!/bin/bash
out=gpurun_out/${1:-driver2}; mkdir -p $out
python -m pytest tests/test_parity_gpu.py tests/test_render_mirror.py tests/test_frame_parallel.py -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
timeout 300 python scripts/soak_streams.py 160 > $out/soak.log 2>&1; echo "soak exit $?" >> $out/status.txt
for cfg in "c3 render pipelined 3" "c3 render threads 3" "c3 render pipelined 3" "c3 render threads 3" "c3 render pipelined 2" "c2 render pipelined 3" "c2 render threads 3"; do
  set -- $cfg
  timeout 300 python bench.py --workload $1 --boundary $2 --driver $3 --streams $4 --steps 240 --warmup 20 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), d['ms_per_step'])" >> $out/rates.txt
done
# host floor: a cloud so small that the GPU work per frame is negligible
for d in pipelined threads; do
  timeout 120 python bench.py --workload c2 --gaussians 2000 --driver $d --streams 3 --steps 600 --warmup 30 --no-cpu-baseline --no-reference-hip 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hostfloor $d', round(d['value'],1), d['ms_per_step'])" >> $out/rates.txt
done
cat $out/status.txt $out/rates.txt; tail -4 $out/pytest.log; tail -4 $out/soak.log
