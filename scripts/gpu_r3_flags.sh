#!/bin/bash
# same-box A/B of bench.py flag sets (alternating, 3 rounds).  usage: gpu_r3_flags.sh tag "wl|flags" "wl|flags" ...
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; rm -f $OUT/ab.txt
for rep in 1 2 3; do
for spec in "$@"; do
wl=${spec%%|*}; fl=${spec#*|}
timeout 300 python bench.py --workload $wl --regions 3 --no-cpu-baseline --no-reference-hip --no-also $fl 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '[$fl]', round(d['value'],1), d['value_serial'], d['roofline']['frame']['single_stream_ms_p50'], ' '.join('%s=%.4f' % (k, v['ms']) for k, v in d['roofline']['stages'].items()))" >> $OUT/ab.txt
done; done
python - <<PY
import collections, re
acc = collections.defaultdict(list)
for l in open("$OUT/ab.txt"):
    m = re.match(r"(\S+) (\[.*?\]) (\S+) (\S+) (\S+) (.*)", l)
    acc[(m.group(1), m.group(2))].append((float(m.group(3)), float(m.group(4)), float(m.group(5)), m.group(6)))
for (wl, fl), v in sorted(acc.items()):
    print(f"{wl:6s} {fl:34s} fps {sorted(x[0] for x in v)}  serial fps {sorted(x[1] for x in v)}  1-stream ms {sorted(x[2] for x in v)[len(v)//2]}")
    print("        ", v[len(v) // 2][3])
PY
