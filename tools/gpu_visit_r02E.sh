#!/bin/bash
# Round 2, visit E: whole GPU suite + smoke on the final tree of the round.
set -u
TAG=${1:-r02E}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -1 $OUT/smoke.log
