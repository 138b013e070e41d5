#!/bin/bash
# Round 6: rocprofv3 kernel stats of the frame loop (C2 + objects, 4 files per frame), both PNG modes.
out=gpurun_out/${1:-r6e}; mkdir -p $out; export TMPDIR=/tmp
for mode in 1 0; do
  ( cd /tmp && GSR_PNG_DEFLATE=$mode AUTOVFX_AMD_LOOP_STREAMS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof$mode" -o run -- \
      python "$GRAFT_REPO_ROOT/scripts/bench_loop.py" --frames 200 --reference-frames 1 > "$GRAFT_REPO_ROOT/$out/loop$mode.json" 2> "$GRAFT_REPO_ROOT/$out/prof$mode.err" )
  F=$(find "$out/prof$mode" -name "*kernel_stats.csv" | head -1); cp "$F" "$out/kernel_stats_deflate$mode.csv"
  find "$out/prof$mode" -type f -size +8M -delete 2>/dev/null
  echo "== GSR_PNG_DEFLATE=$mode"
  python - "$out/kernel_stats_deflate$mode.csv" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:32]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name'])
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['TotalDurationNs'])/1e6:9.2f}ms {float(r['Percentage']):6.2f}%  {n[:90]}")
PY
done
