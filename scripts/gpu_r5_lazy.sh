#!/bin/bash
out=gpurun_out/${1:-r5k}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_raw_autograd_gpu.py tests/test_raw_gpu.py tests/test_render_mirror.py tests/test_compositor.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -5 $out/pytest.log
timeout 600 python - > $out/train_render.txt 2>&1 <<'PY'
import json, sys, torch
sys.path.insert(0, '.')
import bench
from autovfx_amd import renderer
dev = torch.device('cuda', 0)
for rep in range(2):
    for lazy in (False, True):
        renderer.LAZY_NORMAL_GRADIENTS = lazy
        r = bench.training_render_iteration(dev)
        print("lazy", lazy, json.dumps({k: r[k] for k in r if k in ("ms_per_iter", "reference_structure_ms_per_iter")}), flush=True)
PY
cat $out/status.txt $out/train_render.txt
