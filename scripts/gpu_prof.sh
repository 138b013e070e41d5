#!/bin/bash
# rocprofv3 kernel trace of one bench configuration; prints the per-kernel summary.  usage: gpu_prof.sh tag "bench args"
set -u
TAG=$1; ARGS=${2:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o run -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --streams 1 $ARGS > "$GRAFT_REPO_ROOT/$OUT/bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
echo "rocprof exit $?"
find "$OUT/prof" -type f -size +8M -delete 2>/dev/null
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/kernel_stats.csv"
python - "$OUT/kernel_stats.csv" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r'rocprim::ROCPRIM_\d+_NS::','',r['Name'])
    n=re.sub(r'\(anonymous namespace\)::','',n)
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['Percentage']):6.2f}%  {n[:110]}")
PY
