#!/bin/bash
# Round 2, visit e: engine-vs-Python-plan bitwise tests (host-side weight-norm fold in both plans), style path on the
# HIP kernels (ABI v10), host issue-time breakdown.
set -u
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest engine + style"; timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_style_gpu.py -m gpu -q --maxfail=30 > $OUT/pytest_engine_style.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_engine_style.log; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $OUT/pytest_engine_style.log | head -40
echo "== probe host"; timeout 600 python tools/probe_host.py > $OUT/probe_host.log 2>&1; echo "exit $?"; head -75 $OUT/probe_host.log
