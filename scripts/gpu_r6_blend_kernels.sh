#!/bin/bash
# Round 6: blend_frames -- one layer from file to GPU memory by stage (bench_layer_io.py), the bench leg at 8 / 16 / 32 pool threads,
# and rocprofv3 kernel stats of the leg.
out=gpurun_out/${1:-r6blend}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python scripts/bench_layer_io.py > $out/layer_io.json 2> $out/layer_io.err; echo "layer_io exit $?" > $out/status.txt
cat $out/layer_io.json
for t in 8 16 32; do
  AUTOVFX_AMD_BLEND_STATS=1 AUTOVFX_AMD_BLEND_DECODERS=$t timeout 400 python scripts/bench_loop.py --legs c5_blend_frames --frames 400 > $out/blend_$t.json 2> $out/blend_$t.err
  echo "blend $t exit $?" >> $out/status.txt; tail -1 $out/blend_$t.json
done
( cd /tmp && AUTOVFX_AMD_BLEND_DECODERS=16 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o run -- \
    python "$GRAFT_REPO_ROOT/scripts/bench_loop.py" --legs c5_blend_frames --frames 200 > "$GRAFT_REPO_ROOT/$out/blend_prof.json" 2> "$GRAFT_REPO_ROOT/$out/prof.err" )
F=$(find "$out/prof" -name "*kernel_stats.csv" | head -1); cp "$F" "$out/blend_kernel_stats.csv"
find "$out/prof" -type f -size +8M -delete 2>/dev/null
python - "$out/blend_kernel_stats.csv" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:24]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name'])
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['TotalDurationNs'])/1e6:9.2f}ms {float(r['Percentage']):6.2f}%  {n[:90]}")
PY
cat $out/status.txt
