#!/bin/bash
# Exercise bench.py's N > 1 code path (process group, barriers, pipelined RCCL gather) with a one-rank group.
out=gpurun_out/${1:-dist1}; mkdir -p $out
timeout 300 python bench.py --force-distributed --no-cpu-baseline --no-reference-hip > $out/a.json 2> $out/a.err; echo "plain exit $?" > $out/status.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 10 --force-distributed --no-cpu-baseline --no-reference-hip > $out/b.json 2> $out/b.err; echo "torchrun exit $?" >> $out/status.txt
timeout 300 python bench.py --force-distributed --boundary render --driver threads --no-cpu-baseline --no-reference-hip > $out/c.json 2> $out/c.err; echo "render/threads exit $?" >> $out/status.txt
cat $out/status.txt
for f in a b c; do python -c "import sys,json; d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'].get('gathered_frames'), d['config'].get('stream_driver'))" || tail -5 $out/$f.err; done
