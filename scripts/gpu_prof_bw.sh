#!/bin/bash
# rocprofv3 kernel trace of the forward+backward iteration bench.  usage: gpu_prof_bw.sh tag [workload]
set -u
TAG=$1; WL=${2:-c3}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o run -- \
    python "$GRAFT_REPO_ROOT/scripts/bench_backward.py" --workload $WL > "$GRAFT_REPO_ROOT/$OUT/bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
echo "rocprof exit $?"; cat "$OUT/bench.json"
find "$OUT/prof" -type f -size +8M -delete 2>/dev/null
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/kernel_stats.csv"
python - "$OUT/kernel_stats.csv" <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    n=re.sub(r'rocprim::ROCPRIM_\d+_NS::','',r['Name'])
    n=re.sub(r'\(anonymous namespace\)::','',n)
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['Percentage']):6.2f}%  {n[:120]}")
PY
