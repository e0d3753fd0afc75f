#!/bin/bash
# Round 2, visit d: engine-vs-Python-plan bitwise tests + decoder graph capture, sub-batch (Infinity Cache) probe,
# host issue time per step for both plans.
set -u
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest test_engine_gpu"; timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --maxfail=30 > $OUT/pytest_engine.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_engine.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_engine.log | head -30
echo "== probe subbatch"; timeout 600 python tools/probe_subbatch.py > $OUT/probe_subbatch.log 2>&1; echo "exit $?"; tail -20 $OUT/probe_subbatch.log
echo "== bench single (engine plan)"; timeout 600 python bench.py --single-stream --no-cpu-baseline --steps 5 > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "import json;r=json.load(open('$OUT/bench_single.json'));print(r['ms_per_step'], r['config']['host_issue_ms_per_step'], r['config']['plan'])"
echo "== bench single (python plan)"; ST2_PLAN=python timeout 600 python bench.py --single-stream --no-cpu-baseline --steps 5 > $OUT/bench_single_pyplan.json 2> $OUT/bench_single_pyplan.err; python -c "import json;r=json.load(open('$OUT/bench_single_pyplan.json'));print(r['ms_per_step'], r['config']['host_issue_ms_per_step'], r['config']['plan'])"
