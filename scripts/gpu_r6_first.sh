#!/bin/bash
# Round 6, first GPU call: the new frame-loop tests, the tests of what they lean on, the end-to-end loop leg.
out=gpurun_out/${1:-r6a}; mkdir -p $out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_frame_loop.py tests/test_frame_io.py tests/test_dynamic_scene.py -x -q -m gpu -p no:cacheprovider ) > $out/pytest.log 2>&1; echo "pytest exit $?" > $out/status.txt
tail -15 $out/pytest.log
timeout 600 python scripts/bench_loop.py --frames 400 > $out/c5_loop.json 2> $out/c5_loop.err; echo "loop exit $?" >> $out/status.txt
cat $out/status.txt; tail -c 1500 $out/c5_loop.json; tail -5 $out/c5_loop.err
