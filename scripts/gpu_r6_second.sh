#!/bin/bash
# Round 6, second GPU call: frames for offline encoder work; streams / writer-threads sweep of the frame loop.
out=gpurun_out/${1:-r6b}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python scripts/dump_frames.py $out > $out/dump.log 2>&1; echo "dump exit $?" > $out/status.txt
for s in 3 5 7; do for w in 4 8; do
  AUTOVFX_AMD_LOOP_STREAMS=$s AUTOVFX_AMD_LOOP_WRITERS=$w timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 2 > $out/loop_s${s}_w${w}.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_s${s}_w${w}.json"))["c5_loop"]
print("streams $s writers $w:", d["value"], "frames/s", d["ms_per_frame"], "ms")
PY
done; done | tee $out/sweep.txt
cat $out/status.txt; tail -3 $out/dump.log
