#!/bin/bash
# parity tests + a few bench lines after a change.  usage: scripts/gpu_tests_bench.sh tag
set -u
TAG=${1:-tb}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -25 > $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
cp -f gpurun_out/parity_report.jsonl $OUT/ 2>/dev/null
grep -h "heavy" $OUT/parity_report.jsonl | grep -E "size|:cull|inference" | cut -c1-220
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 3 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], d['roofline']['frame']['single_stream_ms_p50'], {k:v['ms'] for k,v in d['roofline']['stages'].items()}, d['roofline']['slab_pairs_last_frame'])"; }
run c3
run c2 --workload c2
run c2_nodefer --workload c2 --no-defer-colour
run heavy --workload heavy
run heavy_s2 --workload heavy --slabs 2
run heavy_s3 --workload heavy --slabs 3
run heavy1080 --workload heavy1080
run heavy1080_s2 --workload heavy1080 --slabs 2
run heavy1080_s3 --workload heavy1080 --slabs 3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_heavy" -o h -- \
    python "$GRAFT_REPO_ROOT/bench.py" --workload heavy --steps 20 --warmup 5 --profile-run --streams 1 > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/prof_heavy.err" )
find $OUT/prof_heavy -type f -size +8M -delete
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_heavy/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        n = r['Name'].replace('gsr::(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
        print(f"{n:42s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']} %")
PY
