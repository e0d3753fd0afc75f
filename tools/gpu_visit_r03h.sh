#!/bin/bash
# Round 3, visit h: BiLSTM with packed fmas (probe_lstm: us / step, block size 4 vs 8), whole GPU suite, default bench.
set -u
TAG=${1:-r03h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== probe_lstm"; timeout 200 python tools/probe_lstm.py 2>&1 | tee $OUT/probe_lstm.log | grep -v Warn
echo "== bench"; timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])"
echo "== pytest -m gpu"; timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
