#!/bin/bash
out=gpurun_out/${1:-r5p}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_io.py tests/test_dynamic_scene.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -4 $out/pytest.log
for s in 1 3 5 7; do
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 2000 960x540 --streams $s --out /dev/shm/r5p_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5p_traj
done
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 2000 960x540 --streams 5 --writer-threads 8 --out /dev/shm/r5p_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5p_traj
cat $out/status.txt $out/traj.json; tail -3 $out/traj.err
