#!/bin/bash
# PMC passes over the training iteration (scripts/bench_backward.py), reduced to the two backward kernels.
# usage: gpu_pmc_bw.sh tag [workload]
set -u
TAG=${1:-pmcbw}; WL=${2:-c3}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/scripts/bench_backward.py --workload $WL --steps 6 > $OUT/$name.json 2> $OUT/$name.err
  echo "$name exit $?"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run mem FETCH_SIZE WRITE_SIZE
run ea TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("gsr::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "backward" in n:
            a = acc[(n, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$OUT/reduced.txt", "w") as out:
    for (n, c), (t, k) in sorted(acc.items()):
        line = f"{n:48s} {c:40s} {t / k:16.1f} x{k}"
        print(line); out.write(line + "\n")
PY
find $OUT -name "*.csv" -size +4M -delete
