#!/bin/bash
# Round 2, visit z: validation of the round-2 tree (shared epilogue, fused kernel for HBM-bound layers, split-K, graphed front):
# whole GPU suite, smoke, bench for every config, rocprofv3 stats.
set -u
TAG=${1:-r02z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -30
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -1 $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['cpu_baseline'])"; tail -2 $OUT/bench.err
echo "== bench single"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "import json;r=json.load(open('$OUT/bench_single.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['avg_launch_ms']); [print(c['ks'],c['C_in'],c['L'],c['avg_launch_ms'],c['frac'],c['share']) for c in r['roofline']['classes']]"
for c in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['config'].get('first_chunk_latency_ms'))"
done
echo "== rocprof stats single-stream"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o bench1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof1_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_single_kernel_stats.csv; done
head -12 $OUT/bench_single_kernel_stats.csv 2>/dev/null | cut -c1-160
