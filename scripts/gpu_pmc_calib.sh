#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (and the request-size-resolved L2 <-> fabric counters) on known byte counts.
# usage: scripts/gpu_pmc_calib.sh tag
set -u
TAG=${1:-calib}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/scripts/ubench/pmc_calib.hip -o /tmp/pmc_calib || exit 1
timeout 120 /tmp/pmc_calib > $OUT/table.csv; echo "plain run exit $?"
pass() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- /tmp/pmc_calib > /dev/null 2> $OUT/$name.err; echo "$name exit $?"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdsize TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass dram TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_BUBBLE_sum
pass wrsize TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python $GRAFT_REPO_ROOT/scripts/pmc_calib_reduce.py $OUT $OUT/pmc_calibration.csv
find $OUT -name "*kernel_trace.csv" -delete
