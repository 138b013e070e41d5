#!/bin/bash
# The N > 1 code paths of bench.py with a one-rank RCCL group (process group, barriers, pipelined gather; weak and strong
# scaling modes), plain and under torch.distributed.run.  usage: scripts/gpu_dist1.sh tag
set -u
TAG=${1:-dist1}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
show() { python -c "import json; d=json.load(open('$1')); print('$1', d['value'], d['scaling'], d['regions']['values'], d['config'].get('gather'), d['config'].get('job_frames'))"; }
timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 3 > $OUT/n1.json 2> $OUT/n1.err; show $OUT/n1.json
timeout 300 python bench.py --force-distributed --no-cpu-baseline --no-reference-hip --no-also --regions 3 > $OUT/weak.json 2> $OUT/weak.err; echo "weak exit $?"; show $OUT/weak.json
timeout 300 python bench.py --force-distributed --job-frames 801 --no-cpu-baseline --no-reference-hip --no-also --regions 3 > $OUT/strong.json 2> $OUT/strong.err; echo "strong exit $?"; show $OUT/strong.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-distributed --job-frames 800 --steps 50 --warmup 5 --no-cpu-baseline --no-reference-hip --no-also --regions 2 > $OUT/strong_torchrun.json 2> $OUT/strong_torchrun.err; echo "torchrun exit $?"; show $OUT/strong_torchrun.json
for f in $OUT/*.err; do echo "== $f"; tail -n 3 $f; done | grep -v amdgpu.ids | head -40
