#!/bin/bash
# Same-box A/B of two TREES (the working tree and a git worktree of another commit under .ab_base/, built there):
# alternating short bench runs, headline and serial rate of each.  usage: scripts/gpu_ab_tree.sh [rounds] [extra bench flags]
set -u
R=${1:-3}; shift || true
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 $R); do
  for t in base new; do
    d=$([ $t = base ] && echo .ab_base || echo .)
    ( cd $d && timeout 300 python bench.py --steps 40 --warmup 10 --regions 5 --no-cpu-baseline --no-also --no-reference-hip "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline']['stages']; print('$t', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], {k:v['ms'] for k,v in st.items()})" )
  done
done
