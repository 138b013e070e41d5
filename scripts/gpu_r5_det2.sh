#!/bin/bash
out=gpurun_out/${1:-r5s}; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_cabi_native.py tests/test_raw_autograd_gpu.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -3 $out/pytest.log
for w in c2 c3; do GSR_BACKWARD_DETERMINISTIC=1 timeout 200 python scripts/bench_backward.py --workload $w --steps 30 2>/dev/null | tail -1 >> $out/det.txt; done
( cd /tmp && GSR_BACKWARD_DETERMINISTIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o det -- python $GRAFT_REPO_ROOT/scripts/bench_backward.py --workload c3 --steps 10 > /dev/null 2>&1 )
F=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $out/det_c3_kernel_stats.csv; rm -rf $out/prof
cat $out/status.txt $out/det.txt; grep -E "det_reduce|render_backward|det_segments" $out/det_c3_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120
