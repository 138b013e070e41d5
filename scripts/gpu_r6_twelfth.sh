#!/bin/bash
# Round 6: the gradient tests with the per-element bar and the option guard; frame-file kernels in both modes.
out=gpurun_out/${1:-r6u}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1800 python -m pytest tests/test_backward_gpu.py tests/test_raw_autograd_gpu.py tests/test_reference_hip_gpu.py -q -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
tail -15 $out/pytest.log
python - <<'PY'
import json
worst={}
for line in open("gpurun_out/parity_report.jsonl"):
    d=json.loads(line)
    if not d["test"].startswith("grad:"): continue
    for k,v in d.items():
        if k.endswith(".frac_elem"):
            if v > worst.get(k,(0,""))[0]: worst[k]=(v,d["test"])
print("worst per-element fractions:", {k:(round(v,3),t) for k,(v,t) in worst.items()})
PY
python scripts/bench_frame_files.py > $out/frameio_kernels.jsonl 2> $out/frameio.err; echo "frameio exit $?" >> $out/status.txt
cut -c1-330 $out/frameio_kernels.jsonl; cat $out/status.txt
