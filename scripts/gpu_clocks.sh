#!/bin/bash
# GPU clocks / power while the C++ frame driver runs (is the blend clock- or power-limited?)
out=gpurun_out/${1:-clocks}; mkdir -p $out
python - <<PY
import sys; sys.path.insert(0,"tests")
import torch, test_cabi_native as t
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
cloud = scenes.config_c3(); cam = orbit_cameras(800,1920,1080)[10]
t.write_scene("/tmp/c3.bin", cloud, cam, torch.zeros(3), 1920, 1080)
PY
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6 > $out/idle.txt
( examples/bin/render_stream /tmp/c3.bin /tmp/o.bin 6000 3 2> $out/run.txt ) &
pid=$!
sleep 1.0
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6 >> $out/busy.txt; sleep 0.5; done
wait $pid
cat $out/idle.txt; echo ---; cat $out/busy.txt; cat $out/run.txt
