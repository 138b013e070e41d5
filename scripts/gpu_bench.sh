#!/bin/bash
# The default bench line (as the driver runs it) + parity tests.  usage: scripts/gpu_bench.sh tag [bench args]
set -u
TAG=${1:-bench}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? in $(( $(date +%s) - t0 )) s"
tail -5 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(d["value"], d["regions"], d["roofline"]["frame"])
print({k: v["ms"] for k, v in d["roofline"]["stages"].items()})
print({k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "launches_per_frame", "avg_launch_ms")})
cb = d.get("cpu_baseline") or {}
print({k: cb.get(k) for k in ("value", "parity", "torch_cpu_c1", "torch_cpu_c2", "torch_cpu_c3")})
print(d.get("reference_on_gpu"))
for k, v in (d.get("also") or {}).items():
    print(k, json.dumps(v)[:600])
PY
