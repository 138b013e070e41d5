#!/bin/bash
# Round 6: the frame loop with the filer thread: tests, then streams sweep in both PNG modes.
out=gpurun_out/${1:-r6i}; mkdir -p $out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_frame_loop.py tests/test_frame_io.py -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -5 $out/pytest.log
for s in 3 5 7 11; do for mode in 1 0; do
  AUTOVFX_AMD_LOOP_STATS=1 GSR_PNG_DEFLATE=$mode AUTOVFX_AMD_LOOP_STREAMS=$s timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_s${s}_d$mode.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_s${s}_d$mode.json"))["c5_loop"]
print("streams $s deflate $mode:", d["value"], "frames/s", d["ms_per_frame"], "ms", d["bytes_per_frame"], d.get("host_seconds"))
PY
done; done | tee $out/sweep.txt
cat $out/status.txt; tail -3 $out/loop.err
