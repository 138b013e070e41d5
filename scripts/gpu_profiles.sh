#!/bin/bash
# The profile sets kept under profiles/: kernel stats of the bench command (product only), kernel stats of the
# forward+backward iteration, PMC passes reduced to per-kernel means and the frame's HBM traffic.
# usage: gpu_profiles.sh tag [commit the tree was taken at]
set -u
TAG=${1:-profiles}; COMMIT=${2:-unknown}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof" -o c3 -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --profile-run --streams 1 > "$GRAFT_REPO_ROOT/$OUT/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/prof.err" )
echo "rocprof bench exit $?"
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/c3_kernel_stats.csv
find $OUT/prof -type f -size +8M -delete
bash scripts/gpu_prof_bw.sh $TAG/bw c3 > $OUT/bw.txt 2>&1; echo "rocprof backward exit $?"
bash scripts/gpu_pmc.sh $TAG/pmc > $OUT/pmc.txt 2>&1; echo "pmc exit $?"
python scripts/pmc_reduce.py $OUT/pmc $OUT/pmc_per_kernel_mean.csv $OUT/traffic.json $COMMIT > /dev/null
head -12 $OUT/c3_kernel_stats.csv | cut -c1-100; grep "blend_quadrant" $OUT/pmc_per_kernel_mean.csv; grep frame_bytes $OUT/traffic.json
