#!/bin/bash
# Quick A/B: parity tests + bench variants.  usage: scripts/gpu_ab.sh tag "args1" "args2" ...
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -30 > "$OUT/pytest_gpu.log"
echo "pytest exit ${PIPESTATUS[0]}" | tee "$OUT/status.txt"
cp -f gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null
i=0
for a in "$@"; do
  timeout 600 python bench.py --no-cpu-baseline $a > "$OUT/bench_$i.json" 2> "$OUT/bench_$i.err"
  echo "bench[$a] exit $?" | tee -a "$OUT/status.txt"
  python - "$OUT/bench_$i.json" "$a" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d["roofline"]["stages"]
    print(sys.argv[2], "| fps", d["value"], "ms", d["ms_per_step"], "|", " ".join(f"{k}={v['ms']:.3f}" for k,v in st.items()))
except Exception as e:
    print("parse fail", e)
PY
  i=$((i+1))
done
tail -4 "$OUT/pytest_gpu.log"
