#!/bin/bash
# Round 6, third GPU call: the deflate PNG encoder (tests, then the loop in both modes).
out=gpurun_out/${1:-r6c}; mkdir -p $out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_frame_io.py -x -q -m gpu -p no:cacheprovider -k "deflate or png_modes" -s ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -40 $out/pytest.log
for mode in 1 0; do
  GSR_PNG_DEFLATE=$mode AUTOVFX_AMD_LOOP_STREAMS=5 timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 2 > $out/loop_deflate$mode.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_deflate$mode.json"))["c5_loop"]
print("GSR_PNG_DEFLATE=$mode:", d["value"], "frames/s", d["ms_per_frame"], "ms", d["bytes_per_frame"], "bytes/frame", d["png"])
PY
done | tee $out/modes.txt
cat $out/status.txt; tail -5 $out/loop.err
