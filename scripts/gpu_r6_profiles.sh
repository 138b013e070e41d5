#!/bin/bash
# Round 6: the profile set of profiles/r06_* on the final tree (kernel stats of the bench command and of the training iteration,
# PMC passes reduced per kernel, the frame's HBM traffic), then the loop's kernel stats in both PNG modes.
out=gpurun_out/${1:-r6prof}; mkdir -p $out; export TMPDIR=/tmp
bash scripts/gpu_profiles.sh ${1:-r6prof}/prof ${2:-unknown} > $out/profiles.txt 2>&1; echo "profiles exit $?" > $out/status.txt
tail -20 $out/profiles.txt
bash scripts/gpu_r6_loop_kernels.sh ${1:-r6prof}/loop > $out/loop_kernels.txt 2>&1; echo "loop kernels exit $?" >> $out/status.txt
grep -A14 "GSR_PNG_DEFLATE=1" $out/loop_kernels.txt | head -20
python scripts/bench_frame_files.py > $out/frameio_kernels.jsonl 2> $out/frameio.err; echo "frameio exit $?" >> $out/status.txt
cat $out/status.txt
