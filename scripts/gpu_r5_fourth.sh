#!/bin/bash
out=gpurun_out/${1:-r5d}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_io.py tests/test_compositor.py tests/test_parity_gpu.py -q -m gpu -p no:cacheprovider -k "png or frame_writer or resize or blender_resolution or unfused or compositor" > $out/pytest_new.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -12 $out/pytest_new.log
for n in 400 2000; do
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit $n 960x540 --out /dev/shm/r5d_traj >> $out/traj.json 2>> $out/traj.err; echo "traj exit $?" >> $out/status.txt
du -sh /dev/shm/r5d_traj 2>/dev/null | tail -1 >> $out/traj.json; rm -rf /dev/shm/r5d_traj
done
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 1000 960x540 --writer-threads 8 --out /dev/shm/r5d_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5d_traj
timeout 300 python -c "
import cProfile, pstats, torch, sys
sys.argv=['render_trajectory.py','--synthetic','1000000','--orbit','600','960x540','--out','/dev/shm/r5d_prof']
sys.path.insert(0,'scripts')
import runpy
cProfile.run('runpy.run_path(\"scripts/render_trajectory.py\", run_name=\"__main__\")', '/tmp/prof.out')
pstats.Stats('/tmp/prof.out').sort_stats('cumulative').print_stats(35)
" > $out/traj_profile.txt 2>&1
rm -rf /dev/shm/r5d_prof
cat $out/status.txt; cat $out/traj.json; df -h /dev/shm | tail -1
