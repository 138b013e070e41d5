#!/bin/bash
# the GPU suite (or the given test paths / -k expression), full tail kept.  usage: gpu_r3_tests.sh tag [pytest args...]
set -u
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf gpurun_out/failures
ARGS=${@:-tests}
timeout 1500 python -X faulthandler -m pytest $ARGS -q -m gpu -p no:cacheprovider -x 2>&1 | tail -80 > $OUT/pytest.log
echo "pytest exit ${PIPESTATUS[0]}"; tail -12 $OUT/pytest.log | cut -c1-400
