#!/bin/bash
# Quick A/B visit: conv probe + conv tests + bench (both stream modes).
set -u
TAG=${1:-r01n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== conv tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv1d" > $OUT/pytest_conv.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_conv.log
echo "== probe conv"; timeout 300 python tools/probe_conv.py > $OUT/probe_conv.log 2>&1; grep "^{" $OUT/probe_conv.log | cut -c1-330
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-230 $OUT/bench.json
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-230 $OUT/bench_single.json
