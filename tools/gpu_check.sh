#!/bin/bash
# One GPU-box visit: parity tests, stage probe, bench line, rocprofv3 kernel stats of the same bench command.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
echo "== smoke" ; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== probe" ; timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e.log 2>&1; tail -8 $OUT/probe_e2e.log
echo "== bench" ; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof" ; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_prof.err ); echo "rocprof exit $?"
find /tmp/prof_$TAG -name '*stats*' -o -name '*domain*' | head; for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/; done
head -30 $OUT/*kernel_stats.csv 2>/dev/null
