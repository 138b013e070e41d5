#!/bin/bash
# same-box A/B of bench.py runs that differ in flags and / or environment, alternating, REPS rounds; parity tests first.
# usage: gpu_r3_flags_env.sh tag "tests ..." "wl|ENV=.. ENV=..|--flags" ...
set -u
TAG=$1; TESTS=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; rm -f $OUT/ab.txt; export TMPDIR=/tmp
if [ -n "$TESTS" ]; then
timeout 1500 python -m pytest $TESTS -q -m gpu -p no:cacheprovider -x 2>&1 | tail -25 > $OUT/pytest.log; tail -3 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log && ! grep -q "failed" $OUT/pytest.log || { grep -E "^E  |Error" $OUT/pytest.log | head -20; echo "TESTS FAILED"; }
fi
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
IFS='|' read -r wl ev fl <<< "$spec"
env $ev timeout 300 python bench.py --workload $wl --regions 3 --no-cpu-baseline --no-reference-hip --no-also $fl 2>/dev/null | tail -1 \
  | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '[$ev $fl]', round(d['value'],1), d['value_serial'], d['roofline']['frame']['single_stream_ms_p50'], ' '.join('%s=%.4f' % (k, v['ms']) for k, v in d['roofline']['stages'].items()))" >> $OUT/ab.txt
done; done
python - <<PY
import collections, re
acc = collections.defaultdict(list)
for l in open("$OUT/ab.txt"):
    m = re.match(r"(\S+) (\[.*?\]) (\S+) (\S+) (\S+) (.*)", l)
    acc[(m.group(1), m.group(2))].append((float(m.group(3)), float(m.group(4)) if m.group(4) != 'None' else 0.0, float(m.group(5)), m.group(6)))
for (wl, fl), v in sorted(acc.items()):
    print(f"{wl:6s} {fl:40s} fps {sorted(x[0] for x in v)}  serial fps {sorted(x[1] for x in v)}  1-stream ms {sorted(x[2] for x in v)[len(v)//2]}")
    print("        ", v[len(v) // 2][3])
PY
