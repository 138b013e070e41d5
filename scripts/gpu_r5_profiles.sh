#!/bin/bash
# Round 5: the profile set of profiles/r05_* on the final tree, and the driver's test command once more on this box.
out=gpurun_out/${1:-r5f}; mkdir -p $out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
cp -f gpurun_out/parity_report.jsonl $out/ 2>/dev/null
tail -6 $out/pytest.log
bash scripts/gpu_profiles.sh ${1:-r5f}/prof ${2:-unknown} > $out/profiles.txt 2>&1; echo "profiles exit $?" >> $out/status.txt
tail -20 $out/profiles.txt; cat $out/status.txt
