#!/bin/bash
# quick same-box A/B of the backward kernels (rocprofv3 kernel averages), base = .ab_base
out=gpurun_out/${1:-r6x}; mkdir -p $out; export TMPDIR=/tmp
for t in base new; do
  d=$([ $t = base ] && echo $GRAFT_REPO_ROOT/.ab_base || echo $GRAFT_REPO_ROOT)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof_$t" -o run -- \
      python "$d/scripts/bench_backward.py" --workload ${2:-c3} --steps 20 > "$GRAFT_REPO_ROOT/$out/bw_$t.json" 2> "$GRAFT_REPO_ROOT/$out/prof_$t.err" )
  F=$(find "$out/prof_$t" -name "*kernel_stats.csv" | head -1); cp "$F" "$out/kernel_stats_$t.csv"
  find "$out/prof_$t" -type f -size +4M -delete 2>/dev/null
  echo "== $t"; grep "backward" "$out/kernel_stats_$t.csv" | awk -F'",' '{print $1}' | cut -c1-60 | paste - <(grep "backward" "$out/kernel_stats_$t.csv" | awk -F, '{print $(NF-4)/1000 " us"}')
done
