#!/bin/bash
out=gpurun_out/${1:-r5e}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_io.py -q -m gpu -p no:cacheprovider > $out/pytest_new.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -5 $out/pytest_new.log
for n in 400 2000; do
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit $n 960x540 --out /dev/shm/r5e_traj >> $out/traj.json 2>> $out/traj.err; echo "traj exit $?" >> $out/status.txt
rm -rf /dev/shm/r5e_traj
done
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 2000 960x540 --writer-threads 8 --out /dev/shm/r5e_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5e_traj
timeout 300 python scripts/render_trajectory.py --synthetic 1000000 --orbit 300 960x540 --deflate --out /dev/shm/r5e_traj >> $out/traj.json 2>> $out/traj.err
rm -rf /dev/shm/r5e_traj
timeout 300 python -c "
import cProfile, pstats, torch, sys
sys.argv=['render_trajectory.py','--synthetic','1000000','--orbit','1000','960x540','--out','/dev/shm/r5e_prof']
import runpy
cProfile.run('runpy.run_path(\"scripts/render_trajectory.py\", run_name=\"__main__\")', '/tmp/prof.out')
pstats.Stats('/tmp/prof.out').sort_stats('tottime').print_stats(25)
" > $out/traj_profile.txt 2>&1
rm -rf /dev/shm/r5e_prof
cat $out/status.txt; cat $out/traj.json
