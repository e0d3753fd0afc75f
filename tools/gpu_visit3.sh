#!/bin/bash
# Short GPU visit: pipeline tests, bench with and without the two-stream overlap, rocprofv3 stats of the default bench.
set -u
TAG=${1:-r01k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (pipeline)"; timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/pytest_sel.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_sel.log
echo "== bench (two streams)"; timeout 900 python bench.py --steps 8 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-260 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench --single-stream"; timeout 900 python bench.py --steps 8 --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; echo "exit $?"; cut -c1-260 $OUT/bench_single.json
echo "== rocprof stats"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_prof.json 2> $R/$OUT/bench_prof.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/; done
cut -c1-200 $OUT/bench_prof.json
