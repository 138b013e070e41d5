#!/bin/bash
# First contact of the radix sort with the GPU: its own tests under a short timeout, then A/B bench lines.
set -u
OUT=gpurun_out/radix
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_radix_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -25 > "$OUT/pytest_radix.log"
echo "radix exit ${PIPESTATUS[0]}" | tee "$OUT/status.txt"
timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -p no:cacheprovider -k "sort_impl" 2>&1 | tail -25 > "$OUT/pytest_ab.log"
echo "ab exit ${PIPESTATUS[0]}" | tee -a "$OUT/status.txt"
for a in "--streams 1 --sort-impl 0" "--streams 1 --sort-impl 1" "--sort-impl 0" "--sort-impl 1"; do
  timeout 300 python bench.py --no-cpu-baseline $a > "$OUT/b.json" 2> "$OUT/b.err"
  python - "$OUT/b.json" "$a" <<'PY' | tee -a "$OUT/status.txt"
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d["roofline"]["stages"]
    print(sys.argv[2], "| fps", d["value"], "ms", d["ms_per_step"], "|", " ".join(f"{k}={v['ms']:.3f}" for k,v in st.items()))
except Exception as e:
    print("parse fail", e, open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
tail -8 "$OUT/pytest_radix.log" "$OUT/pytest_ab.log"
