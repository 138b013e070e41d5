#!/bin/bash
# Kernel trace of the multi-stream bench run + overlap analysis (scripts/trace_overlap.py).
out=gpurun_out/${1:-overlap}; mkdir -p $out
export TMPDIR=/tmp
for S in ${2:-3}; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof$S" -o run -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 120 --warmup 10 --streams $S --regions 2 --no-cpu-baseline --no-reference-hip --no-also > "$GRAFT_REPO_ROOT/$out/bench$S.json" 2> "$GRAFT_REPO_ROOT/$out/prof$S.err" )
f=$(find $out/prof$S -name "*kernel_trace.csv" | head -1)
python scripts/trace_overlap.py $f > $out/overlap$S.txt 2>&1
find $out/prof$S -type f -size +8M -delete
cat $out/overlap$S.txt
done
