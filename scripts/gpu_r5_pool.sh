#!/bin/bash
# Round 5: run pool placed by prefix sums (no atomics in bin_gather) -- the new test first, then the same-box A/B against the
# tree before (.ab_base), then the GPU suite.   usage: scripts/gpu_r5_pool.sh <tag>
set -u
cd "$GRAFT_REPO_ROOT"; out=gpurun_out/${1:-pool}; mkdir -p $out
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "run_pool or c1_every or heavy or wild" 2>&1 | tail -5 > $out/new_test.txt
cat $out/new_test.txt
( bash scripts/gpu_ab_tree.sh 3 --workload heavy; echo ---1080; bash scripts/gpu_ab_tree.sh 2 --workload heavy1080; echo ---c3; bash scripts/gpu_ab_tree.sh 2 ) > $out/ab.txt 2>&1
cat $out/ab.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $out/suite.txt
cat $out/suite.txt
