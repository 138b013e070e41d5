#!/bin/bash
# scheduling experiments with the plain-C++ frame driver (scripts/ubench/stream_sched.cpp)
out=gpurun_out/${1:-sched}; mkdir -p $out
python - <<PY
import sys; sys.path.insert(0,"tests")
import torch, test_cabi_native as t
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
cloud = scenes.config_c3(); cam = orbit_cameras(800,1920,1080)[10]
t.write_scene("/tmp/c3.bin", cloud, cam, torch.zeros(3), 1920, 1080)
PY
for rep in 1 2; do
for cfg in "0 3" "1 3" "2 3" "3 3" "4 3" "5 3" "3 4" "1 4" "0 2" "5 2" "4 2"; do
  set -- $cfg
  GSR_SCHED_MODE=$1 scripts/ubench/stream_sched /tmp/c3.bin /tmp/o.bin 300 $2 2>&1 | sed "s/^/mode $1 S $2: /" | cut -c1-150 >> $out/sched.txt
done; done
cat $out/sched.txt
