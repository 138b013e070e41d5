#!/bin/bash
# Round 6: what bounds the frame loop -- file writes, file kernels + copy, or render + host?
out=gpurun_out/${1:-r6p}; mkdir -p $out; export TMPDIR=/tmp
for diag in "" nowrite nofiles; do for s in 3 7; do
  GSR_LOOP_DIAG=$diag AUTOVFX_AMD_LOOP_STATS=1 AUTOVFX_AMD_LOOP_STREAMS=$s timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_${diag:-full}_s$s.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_${diag:-full}_s$s.json"))["c5_loop"]
print("diag '${diag:-full}' streams $s:", d["value"], "frames/s", d["ms_per_frame"], "ms", d.get("host_seconds"))
PY
done; done | tee $out/diag.txt
tail -3 $out/loop.err
