#!/bin/bash
# one PMC pass of the bench command, reduced to the blend / preprocess rows.  usage: gpu_pmc_one.sh tag counters...
set -u
TAG=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --profile-run --streams 1 > $OUT/p.json 2> $OUT/p.err; echo "exit $?"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("gsr::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "blend" in n or "preprocess" in n or "scatter" in n:
            a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (n, c), (t, k) in sorted(acc.items()):
    print(f"{n:40s} {c:24s} {t / k:14.1f} x{k}")
PY
find $OUT -name "*.csv" -size +4M -delete
