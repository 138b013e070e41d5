#!/bin/bash
# Round-end sanity: parity tests, bench line, the N>1 code path with one rank, multi-stream soak.
out=gpurun_out/${1:-final}; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
timeout 300 python bench.py --no-cpu-baseline --no-reference-hip > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/status.txt
timeout 300 python bench.py --force-distributed --no-cpu-baseline --no-reference-hip > $out/dist.json 2> $out/dist.err; echo "dist exit $?" >> $out/status.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-distributed --no-cpu-baseline --no-reference-hip > $out/dist_tr.json 2> $out/dist_tr.err; echo "torchrun exit $?" >> $out/status.txt
timeout 300 python scripts/soak_streams.py 160 > $out/soak.log 2>&1; echo "soak exit $?" >> $out/status.txt
cat $out/status.txt; tail -3 $out/pytest.log; tail -3 $out/soak.log
for f in bench dist dist_tr; do python -c "import json; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['steps'], d['config'].get('gathered_frames'), d['config'].get('gather_chunks'))"; wc -l $out/$f.json; done
