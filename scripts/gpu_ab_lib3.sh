#!/bin/bash
# A/B two builds of the library on the same box, 3 streams (the bench default), alternating, C3 / C2 / heavy.
# usage: gpu_ab_lib3.sh tag libA libB [workloads...]
out=gpurun_out/${1:-ablib3}; mkdir -p $out; A=$2; B=$3; shift 3; WLS=${@:-c3 c2 heavy}
for wl in $WLS; do
for rep in 1 2 3; do
for lib in "$A" "$B"; do
GSR_LIB=$PWD/$lib timeout 300 python bench.py --workload $wl --regions 5 --no-cpu-baseline --no-reference-hip --no-also 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$lib', round(d['value'],1), d['roofline']['frame']['single_stream_ms_p50'])" >> $out/ab.txt
done; done; done
python - <<PY
import collections
acc = collections.defaultdict(list)
for l in open("$out/ab.txt"):
    wl, lib, v, ss = l.split()
    acc[(wl, lib)].append((float(v), float(ss)))
for (wl, lib), v in sorted(acc.items()):
    print(f"{wl:6s} {lib:40s} fps {sorted(x[0] for x in v)}  single-stream ms {sorted(x[1] for x in v)[len(v)//2]}")
PY
