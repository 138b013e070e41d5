#!/bin/bash
# Same-box A/B of bench.py flag sets (box-to-box spread is +-2 %: only same-box comparisons count), alternating, 3 rounds.
# usage: gpu_ab_flags.sh tag workload "flags A" "flags B" ["flags C" ...]
out=gpurun_out/${1:-abflags}; mkdir -p $out; wl=$2; shift 2
for rep in 1 2 3; do
i=0
for flags in "$@"; do
timeout 300 python bench.py --workload $wl --regions 5 --no-cpu-baseline --no-reference-hip --no-also $flags 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$i', round(d['value'],1), d['roofline']['frame']['single_stream_ms_p50'])" >> $out/ab_$wl.txt
i=$((i+1))
done; done
python - "$@" <<PY
import collections, sys
acc = collections.defaultdict(list)
for l in open("$out/ab_$wl.txt"):
    i, v, ss = l.split()
    acc[int(i)].append((float(v), float(ss)))
for i, v in sorted(acc.items()):
    print(f"$wl [{sys.argv[1 + i]:28s}] fps {sorted(x[0] for x in v)}  single-stream ms {sorted(x[1] for x in v)[len(v)//2]}")
PY
