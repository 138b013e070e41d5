#!/bin/bash
# Round 6: compositor tests with real EXR files; blend_frames end to end; PNG kernel times after the finalisation fix.
out=gpurun_out/${1:-r6k}; mkdir -p $out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_compositor.py tests/test_frame_io.py tests/test_frame_loop.py -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -5 $out/pytest.log
timeout 900 python scripts/bench_loop.py --legs c5_blend_frames --frames 400 > $out/blend_frames.json 2> $out/blend.err; echo "blend exit $?" >> $out/status.txt
tail -c 1500 $out/blend_frames.json; tail -3 $out/blend.err
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
