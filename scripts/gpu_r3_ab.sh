#!/bin/bash
# Round 3 same-box A/B of two library builds: parity tests of the new one first, then alternating bench runs
# (default streams and --streams 1) on C3 / C2 / heavy.  usage: gpu_r3_ab.sh tag libA libB [tests...]
set -u
TAG=$1; A=$2; B=$3; shift 3; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
TESTS=${@:-tests/test_radix_gpu.py tests/test_parity_gpu.py tests/test_poison_gpu.py tests/test_backward_gpu.py}
timeout 900 python -m pytest $TESTS -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > $OUT/pytest.log; tail -3 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log && ! grep -q "failed" $OUT/pytest.log || { grep -E "^E  |Error" $OUT/pytest.log | head; echo "TESTS FAILED"; }
rm -f $OUT/ab.txt
for rep in 1 2 3; do
for wl in c3 c2 heavy; do
for lib in "$A" "$B"; do
for st in 11 1; do
GSR_LIB=$PWD/$lib timeout 300 python bench.py --workload $wl --regions 3 --streams $st --no-cpu-baseline --no-reference-hip --no-also 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$lib', $st, round(d['value'],1), d['value_serial'], d['roofline']['frame']['single_stream_ms_p50'], ' '.join('%s=%.4f' % (k, v['ms']) for k, v in d['roofline']['stages'].items()))" >> $OUT/ab.txt
done; done; done; done
python - <<PY
import collections
acc = collections.defaultdict(list)
for l in open("$OUT/ab.txt"):
    f = l.split()
    acc[(f[0], f[2], f[1])].append((float(f[3]), float(f[4]), float(f[5]), " ".join(f[6:])))
for (wl, st, lib), v in sorted(acc.items()):
    print(f"{wl:6s} streams {st:>2s} {lib:44s} fps {sorted(x[0] for x in v)}  serial fps {sorted(x[1] for x in v)}  1-stream ms {sorted(x[2] for x in v)[len(v)//2]}")
    print("        ", v[len(v) // 2][3])
PY
