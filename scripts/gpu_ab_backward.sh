#!/bin/bash
# Same-box A/B of the training iteration (forward + loss + backward, scripts/bench_backward.py) of two trees: the working tree
# and the git worktree under .ab_base/ (see gpu_ab_tree.sh).  usage: scripts/gpu_ab_backward.sh [rounds] [workload]
set -u
R=${1:-3}; WL=${2:-c3}
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 $R); do
  for t in base new; do
    d=$([ $t = base ] && echo .ab_base || echo .)
    ( cd $d && echo -n "$t " && timeout 300 python scripts/bench_backward.py --workload $WL --steps 20 2>/dev/null | tail -1 )
  done
done
