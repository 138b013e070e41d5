#!/bin/bash
# Same-box A/B of the radix sort's in-wave rank (GSR_OPT_RADIX_RANK): ballots (0) against the verified LDS adds (2, default),
# C3, serial and pipelined; the bench line also carries config.options.radix_rank_fallbacks.  Output: gpurun_out/rank_ab/.
mkdir -p gpurun_out/rank_ab
for rep in 1 2; do
  for mode in 0 2; do
    GSR_RADIX_RANK=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-reference-hip \
      > gpurun_out/rank_ab/mode${mode}_rep${rep}.json 2> gpurun_out/rank_ab/mode${mode}_rep${rep}.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/rank_ab/mode*_rep*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "ms", l["ms_per_step"], "serial ms", l["ms_per_step_serial"], "active", l["config"]["options"]["radix_rank_active"],
              "fallbacks", l["config"]["options"]["radix_rank_fallbacks"])
    except Exception as e:
        print(f, "unreadable:", e)
PY
