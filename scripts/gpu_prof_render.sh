#!/bin/bash
# kernel stats of the render() boundary bench (single stream)
out=gpurun_out/${1:-profrender}; mkdir -p $out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o r -- \
    python "$GRAFT_REPO_ROOT/bench.py" --boundary render --steps 30 --warmup 10 --no-cpu-baseline --no-reference-hip --streams 1 > "$GRAFT_REPO_ROOT/$out/bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof.err" )
find $out/prof -type f -size +8M -delete
python - $(find $out/prof -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n)
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['Percentage']):6.2f}%  {n[:100]}")
PY
python -c "import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
