#!/bin/bash
# Round 3, visit i: cooperative BiLSTM group size (2 / 4 / 8 utterances per group) in isolation and inside the bench.
set -u
TAG=${1:-r03i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== probe_lstm"; timeout 200 python tools/probe_lstm.py 2>&1 | tee $OUT/probe_lstm.log | grep "block\|coop B"
for blk in 0 2 8; do
  echo "== bench --lstm-block $blk"; timeout 400 python bench.py --no-cpu-baseline --lstm-block $blk > $OUT/bench_blk$blk.json 2> $OUT/bench_blk$blk.err
  python -c "import json;r=json.load(open('$OUT/bench_blk$blk.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])"
done
echo "== pytest lstm"; timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k lstm 2>&1 | tail -3
