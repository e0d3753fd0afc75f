#!/bin/bash
# Round 2, visit F: st2_prosody_forward (C++ plan) vs the Python plan, bitwise; engine tests; final default bench line.
set -u
TAG=${1:-r02F}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest engine"; timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_c_host.py -m gpu -q --maxfail=10 > $OUT/pytest_engine.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_engine.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_engine.log | head -20
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'])"
