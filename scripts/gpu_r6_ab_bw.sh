#!/bin/bash
# Round 6: same-box A/B of the training iteration (base = .ab_base worktree, new = working tree), C3, per-kernel stats of both.
out=gpurun_out/${1:-r6m}; mkdir -p $out; export TMPDIR=/tmp
bash scripts/gpu_ab_backward.sh 3 c3 | tee $out/ab_backward_c3.txt
for t in base new; do
  d=$([ $t = base ] && echo $GRAFT_REPO_ROOT/.ab_base || echo $GRAFT_REPO_ROOT)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof_$t" -o run -- \
      python "$d/scripts/bench_backward.py" --workload c3 --steps 20 > "$GRAFT_REPO_ROOT/$out/bw_$t.json" 2> "$GRAFT_REPO_ROOT/$out/prof_$t.err" )
  F=$(find "$out/prof_$t" -name "*kernel_stats.csv" | head -1); cp "$F" "$out/kernel_stats_$t.csv"
  find "$out/prof_$t" -type f -size +4M -delete 2>/dev/null
  echo "== $t"; python - "$out/kernel_stats_$t.csv" <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name'])
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us  {n[:80]}")
PY
done
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_raw_autograd_gpu.py -x -q -p no:cacheprovider > $out/pytest_bw.log 2>&1; tail -3 $out/pytest_bw.log
