#!/bin/bash
# Backward census: lane-efficiency counters (trace build) + rocprofv3 PMC passes over scripts/bench_backward.py at C3.
# usage: scripts/gpu_bw_census.sh [tag]
set -u
TAG=${1:-bw_census}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
GSR_LIB=$GRAFT_REPO_ROOT/autovfx_amd/lib/libgsr_hip_trace.so timeout 300 python "$GRAFT_REPO_ROOT/scripts/backward_census.py" --workload c3 > "$OUT/census_c3.json" 2> "$OUT/census.err"
echo "census exit $?"; cat "$OUT/census_c3.json"
cd /tmp
run() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python "$GRAFT_REPO_ROOT/scripts/bench_backward.py" --workload c3 --steps 6 > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$name exit $?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find "$OUT" -type f -size +6M -delete
GSR_PMC_P=3000000 python "$GRAFT_REPO_ROOT/scripts/pmc_reduce.py" "$OUT" "$OUT/pmc_per_kernel_mean.csv" > /dev/null 2>&1
grep -i "backward" "$OUT/pmc_per_kernel_mean.csv"
