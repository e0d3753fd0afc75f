#!/bin/bash
# Round 2, visit w: split-K only below 128 workgroups: all configurations again.
set -u
TAG=${1:-r02w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
for c in longform libritts_hifigan libritts_istftnet; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"
done
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'])"
echo "== bench again"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench2.json 2> $OUT/bench2.err; python -c "import json;r=json.load(open('$OUT/bench2.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'])"
