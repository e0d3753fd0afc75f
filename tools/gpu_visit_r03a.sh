#!/bin/bash
# Round 3, visit a (prepared at the end of round 2, when no GPU minutes were left): validate the ABI-15 tree as a whole and
# time the C++ front (ST2_FRONT=engine, csrc/st2_engine.hip front_plan) against the Python front it was only
# parity-checked against so far.
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit_r03a.sh r03a'
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
echo "== style plan probe"; timeout 200 python tools/probe_style_plan.py > $OUT/style_plan.log 2>&1; tail -4 $OUT/style_plan.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
echo "== bench (default: Python front under a hipGraph)"; timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'])"
for f in python engine; do   # same box A/B: eager fronts (no front graph) and graphed fronts
  echo "== bench --eager-front ST2_FRONT=$f"; ST2_FRONT=$f timeout 200 python bench.py --eager-front --no-cpu-baseline > $OUT/bench_eager_$f.json 2> $OUT/bench_eager_$f.err
  python -c "import json;r=json.load(open('$OUT/bench_eager_$f.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'])"
  echo "== bench ST2_FRONT=$f (graphed)"; ST2_FRONT=$f timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_graph_$f.json 2> $OUT/bench_graph_$f.err
  python -c "import json;r=json.load(open('$OUT/bench_graph_$f.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'])"
  echo "== bench --config longform ST2_FRONT=$f"; ST2_FRONT=$f timeout 200 python bench.py --config longform --no-cpu-baseline > $OUT/bench_longform_$f.json 2> $OUT/bench_longform_$f.err
  python -c "import json;r=json.load(open('$OUT/bench_longform_$f.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT && find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
