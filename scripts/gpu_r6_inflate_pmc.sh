#!/bin/bash
# Round 6: SQ counters of inflate_zlib_kernel on the inflate ubench (scripts/ubench/inflate_rates.py): instructions by class and wait cycles
# per stream.  Two passes (the counters do not fit one).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6infpmc}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
pass() { tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $GRAFT_REPO_ROOT/scripts/ubench/inflate_rates.py > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag exit $?"; }
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate_zlib" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"]) if r.get("Dispatch_Id") else 0][r["Counter_Name"]] = float(r["Counter_Value"])
# dispatches come in the order of the ubench's cases, four launches each (one warm-up + three timed)
import itertools
keys = sorted(rows)
print(len(keys), "dispatches")
for k in keys:
    print(k, {c: int(v) for c, v in sorted(rows[k].items())})
PY
find $OUT -name "*.csv" -size +4M -delete
