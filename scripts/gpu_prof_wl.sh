#!/bin/bash
# rocprofv3 kernel stats of one bench workload (single stream, uniform calls).  usage: gpu_prof_wl.sh tag workload [bench flags]
set -u
TAG=$1; WL=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_$WL" -o p -- \
    python "$GRAFT_REPO_ROOT/bench.py" --workload $WL --steps 20 --warmup 5 --profile-run --streams 1 "$@" > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/prof_$WL.err" )
find $OUT/prof_$WL -type f -size +8M -delete
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_$WL/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    frames = next(int(r['Calls']) for r in rows if 'preprocess_kernel' in r['Name'])
    tot = 0.0
    for r in rows[:18]:
        n = r['Name'].replace('gsr::(anonymous namespace)::','').replace('void ','').split('(')[0][:40]
        per_frame = float(r['TotalDurationNs']) / frames / 1e3
        tot += per_frame
        print(f"$WL {n:42s} {int(r['Calls'])/frames:5.1f}/frame avg {float(r['AverageNs'])/1e3:8.1f} us  {per_frame:8.1f} us/frame")
    print("$WL total of the listed kernels per frame us", round(tot, 1))
PY
