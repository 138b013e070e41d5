#!/bin/bash
# Round 6: deflate tests after the kernel restructuring; host-time split of the loop at several stream counts; kernel stats.
out=gpurun_out/${1:-r6h}; mkdir -p $out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_frame_io.py -x -q -m gpu -p no:cacheprovider -k "deflate or png_modes" ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -5 $out/pytest.log
for s in 3 7 11; do for mode in 1 0; do
  AUTOVFX_AMD_LOOP_STATS=1 GSR_PNG_DEFLATE=$mode AUTOVFX_AMD_LOOP_STREAMS=$s timeout 300 python scripts/bench_loop.py --frames 400 --reference-frames 1 > $out/loop_s${s}_d$mode.json 2>> $out/loop.err
  python - <<PY
import json
d=json.load(open("$out/loop_s${s}_d$mode.json"))["c5_loop"]
print("streams $s deflate $mode:", d["value"], "frames/s", d["ms_per_frame"], "ms", d["bytes_per_frame"], d.get("host_seconds"))
PY
done; done | tee $out/sweep.txt
( cd /tmp && GSR_PNG_DEFLATE=1 AUTOVFX_AMD_LOOP_STREAMS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o run -- \
      python "$GRAFT_REPO_ROOT/scripts/bench_loop.py" --frames 200 --reference-frames 1 > "$GRAFT_REPO_ROOT/$out/loop_prof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.err" )
F=$(find "$out/prof" -name "*kernel_stats.csv" | head -1); cp "$F" "$out/kernel_stats_deflate.csv"
find "$out/prof" -type f -size +8M -delete 2>/dev/null
python - "$out/kernel_stats_deflate.csv" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name'])
    if 'png' in n or 'copyBuffer' in n or 'preview' in n or 'pack' in n: print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['TotalDurationNs'])/1e6:9.2f}ms  {n[:70]}")
PY
cat $out/status.txt
