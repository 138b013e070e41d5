#!/bin/bash
# Round 3, first visit: the whole GPU suite (with the poisoned-allocation suite, both radix rank forms, the RCCL test),
# the cold-process flake hunt of the first test with and without AMD_SERIALIZE_KERNEL=3, one short bench line.
# usage: scripts/gpu_r3_check.sh tag [flake_runs]
set -u
TAG=${1:-r3a}; RUNS=${2:-10}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl; rm -rf gpurun_out/failures
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 2>&1 | tail -60 > $OUT/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]} in $(( $(date +%s) - t0 )) s" | tee $OUT/status.txt
cp -f gpurun_out/parity_report.jsonl $OUT/ 2>/dev/null
tail -25 $OUT/pytest_gpu.log
for mode in default serialize; do
  fails=0
  for i in $(seq 1 $RUNS); do
    if [ $mode = serialize ]; then export AMD_SERIALIZE_KERNEL=3; else unset AMD_SERIALIZE_KERNEL; fi
    timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "test_backward_sh_scene" > $OUT/run.log 2>&1
    if ! grep -q " passed" $OUT/run.log || grep -q "failed" $OUT/run.log; then
      fails=$((fails+1)); cp $OUT/run.log $OUT/fail_${mode}_$i.log; grep -E "^E  " $OUT/run.log | head -8
    fi
  done
  unset AMD_SERIALIZE_KERNEL
  echo "flake hunt ($mode): runs $RUNS failures $fails" | tee -a $OUT/status.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/status.txt
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("bench", d["value"], d["ms_per_step"], d["roofline"]["frame"]["single_stream_ms_p50"], {k: v["ms"] for k, v in d["roofline"]["stages"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-1500:])
PY
ls gpurun_out/failures 2>/dev/null | head
