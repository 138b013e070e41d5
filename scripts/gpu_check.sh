#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: scripts/gpu_check.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > "$OUT/gpu.txt"
nproc > "$OUT/nproc.txt"
echo "== pytest -m gpu" | tee "$OUT/status.txt"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > "$OUT/pytest_gpu.log"
echo "pytest exit ${PIPESTATUS[0]}" | tee -a "$OUT/status.txt"
cp -f gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null
echo "== smoke" | tee -a "$OUT/status.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke exit $?" | tee -a "$OUT/status.txt"
echo "== bench c3" | tee -a "$OUT/status.txt"
timeout 900 python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
echo "bench exit $?" | tee -a "$OUT/status.txt"
echo "== bench c2" | tee -a "$OUT/status.txt"
timeout 600 python bench.py --workload c2 --no-cpu-baseline --streams 3 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
echo "bench c2 exit $?" | tee -a "$OUT/status.txt"
echo "== rocprofv3 kernel trace" | tee -a "$OUT/status.txt"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o c3 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --profile-run --streams 1 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
echo "rocprof exit $?" | tee -a "$OUT/status.txt"
find "$OUT/prof" -name "*kernel_stats*" -o -name "*_stats.csv" 2>/dev/null | head
# keep only the small summaries (traces can be tens of MB)
find "$OUT/prof" -type f -size +8M -delete 2>/dev/null
cat "$OUT/status.txt"
tail -5 "$OUT/pytest_gpu.log"
cat "$OUT/bench_c3.json"
