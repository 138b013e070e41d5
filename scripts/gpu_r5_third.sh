#!/bin/bash
# Round 5: the new frame-file / resize tests, the deterministic backward under rocprofv3, the bench line with its C5 legs.
out=gpurun_out/${1:-r5c}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_io.py tests/test_compositor.py tests/test_parity_gpu.py -x -q -m gpu -p no:cacheprovider -k "png or frame_writer or resize or blender_resolution or unfused or compositor" > $out/pytest_new.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -12 $out/pytest_new.log
( cd /tmp && GSR_BACKWARD_DETERMINISTIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_det -o det -- python $GRAFT_REPO_ROOT/scripts/bench_backward.py --workload c3 --steps 10 > $GRAFT_REPO_ROOT/$out/prof_det.log 2>&1 )
F=$(find $out/prof_det -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $out/det_c3_kernel_stats.csv
rm -rf $out/prof_det
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench exit $?" >> $out/status.txt
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 200 960x540 --out /dev/shm/r5c_traj > $out/traj.json 2> $out/traj.err; echo "traj exit $?" >> $out/status.txt
du -sh /dev/shm/r5c_traj 2>/dev/null | tail -1 >> $out/traj.json; rm -rf /dev/shm/r5c_traj
cat $out/status.txt; cat $out/traj.json; head -14 $out/det_c3_kernel_stats.csv | cut -c1-160
