#!/bin/bash
# Round 3: the default bench line (new fields) + serial-latency variants.  usage: scripts/gpu_r3_bench.sh tag
set -u
TAG=${1:-r3b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "serial", d["value_serial"], d["ms_per_step_serial"])
print("frame", d["roofline"]["frame"])
for k, v in d["roofline"]["stages"].items(): print(" ", k, v)
for k in ("c3_reference_shaped_render", "backward_c3", "c3_render_boundary", "c2_op"):
    print(k, json.dumps(d.get("also", {}).get(k))[:900])
PY
run() { name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-reference-hip --no-also --regions 3 "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name"
  python -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', d['value'], 'serial', d['value_serial'], d['ms_per_step_serial'], d['roofline']['frame']['single_stream_ms_p50'], {k:v['ms'] for k,v in d['roofline']['stages'].items()}, d['roofline']['slab_pairs_last_frame'])"; }
run c3_s1 --streams 1
run c3_s1_slabs1 --streams 1 --slabs 1
run c3_s1_slabs1_nodefer --streams 1 --slabs 1 --no-defer-colour
run c2_s1 --workload c2 --streams 1
run heavy_s1 --workload heavy --streams 1
run heavy_s1_slabs1 --workload heavy --streams 1 --slabs 1
