#!/bin/bash
out=gpurun_out/${1:-r6q}; mkdir -p $out; export TMPDIR=/tmp
for diag in "" nofiles; do for s in 3 7; do for w in 4 1; do
  GSR_LOOP_DIAG=$diag AUTOVFX_AMD_LOOP_WRITERS=$w AUTOVFX_AMD_LOOP_STATS=1 AUTOVFX_AMD_LOOP_STREAMS=$s timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_${diag:-full}_s${s}_w$w.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_${diag:-full}_s${s}_w$w.json"))["c5_loop"]
print("diag '${diag:-full}' streams $s writers $w:", d["value"], "frames/s", d["ms_per_frame"], "ms", d.get("host_seconds"))
PY
done; done; done | tee $out/diag.txt
tail -3 $out/loop.err
