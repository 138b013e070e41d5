#!/bin/bash
# Quick GPU visit: kernel trace (if the trace library is built), selected tests, a few bench lines.
# usage: scripts/gpu_quick.sh "<pytest -k expr or empty>" "bench args 1" "bench args 2" ...
set -u
OUT=gpurun_out/quick
mkdir -p "$OUT"
K=${1:-}; shift || true
if [ -f autovfx_amd/lib/libgsr_hip_trace.so ]; then
  GSR_LIB=$PWD/autovfx_amd/lib/libgsr_hip_trace.so timeout 300 python scripts/kernel_trace.py --slots ${TRACE_SLOTS:-4} 2>&1 | grep -v amdgpu.ids | tee "$OUT/trace.txt"
fi
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -q -x -m gpu -p no:cacheprovider -k "$K" 2>&1 | tail -15 | tee "$OUT/pytest.log"
fi
for a in "$@"; do
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
