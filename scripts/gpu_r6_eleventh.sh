#!/bin/bash
out=gpurun_out/${1:-r6r}; mkdir -p $out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_frame_io.py tests/test_frame_loop.py -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for s in 3 5 7; do
  AUTOVFX_AMD_LOOP_STATS=1 AUTOVFX_AMD_LOOP_STREAMS=$s timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_s${s}.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_s${s}.json"))["c5_loop"]
print("streams $s:", d["value"], "frames/s", d["ms_per_frame"], "ms; call", d["call_seconds"], "load", d["load_scene_seconds"], d.get("host_seconds"))
PY
done | tee $out/diag.txt
tail -3 $out/loop.err
