#!/bin/bash
# Bench every autovfx_amd/lib/libgsr_var_*.so (kernel variants built by hand) with the same arguments.
set -u
for lib in autovfx_amd/lib/libgsr_var_*.so; do
  GSR_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --streams 1 "$@" > /tmp/b.json 2> /tmp/b.err
  python - /tmp/b.json "$(basename $lib)" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d["roofline"]["stages"]
    print(sys.argv[2], "| fps", d["value"], "ms", d["ms_per_step"], "|", " ".join(f"{k}={v['ms']:.3f}" for k,v in st.items()))
except Exception as e:
    print("parse fail", e, open('/tmp/b.err').read()[-600:])
PY
done
