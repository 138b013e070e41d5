#!/bin/bash
# GPU parity tests only (fast iteration).  usage: scripts/gpu_tests.sh tag [pytest args]
set -u
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x "$@" 2>&1 | tail -70 > "$OUT/pytest_gpu.log"
echo "pytest exit ${PIPESTATUS[0]}"
cp -f gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null
tail -40 "$OUT/pytest_gpu.log"
