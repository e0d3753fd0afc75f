#!/bin/bash
# Full-size parity visit: tests/test_fullsize_gpu.py.
set -u
TAG=${1:-r01q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q > $OUT/pytest_fullsize.log 2>&1; echo "exit $?"; tail -15 $OUT/pytest_fullsize.log
