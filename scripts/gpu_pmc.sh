#!/bin/bash
# PMC counter passes (each in its own run, kernel-trace only) for the bench workload.
# usage: scripts/gpu_pmc.sh [tag]
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
run() { # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --profile-run --streams 1 > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$name exit $?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
find "$OUT" -type f -size +6M -delete
ls -la "$OUT" "$OUT"/*/ | head -50
