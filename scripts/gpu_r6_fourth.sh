#!/bin/bash
# Round 6, fourth GPU call: host profile of the frame loop.
out=gpurun_out/${1:-r6d}; mkdir -p $out; export TMPDIR=/tmp
GSR_LOOP_PROFILE=1 AUTOVFX_AMD_LOOP_STREAMS=5 timeout 400 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_profile.json 2> $out/loop.err
python - <<PY
import json
d=json.load(open("$out/loop_profile.json"))["c5_loop"]
print(d["value"], "frames/s")
print("\n".join(d["host_profile"]))
PY
