#!/bin/bash
out=gpurun_out/${1:-r5n}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_io.py tests/test_compositor.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -4 $out/pytest.log
python scripts/bench_frame_files.py 960x540 > $out/frameio.jsonl 2> $out/frameio.err
python scripts/bench_frame_files.py 1920x1080 >> $out/frameio.jsonl 2>> $out/frameio.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o f -- python $GRAFT_REPO_ROOT/scripts/bench_frame_files.py 960x540 > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err )
F=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $out/frameio_kernel_stats.csv; rm -rf $out/prof
for n in 2000 2000; do
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit $n 960x540 --out /dev/shm/r5n_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5n_traj
done
cat $out/status.txt; cut -c1-200 $out/frameio.jsonl; cat $out/traj.json
python - <<PY
import csv,re
for r in list(csv.DictReader(open("$out/frameio_kernel_stats.csv")))[:12]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n)
    print(f"{int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.1f}us {float(r['Percentage']):6.2f}%  {n[:90]}")
PY
