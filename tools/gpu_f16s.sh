#!/bin/bash
# GPU visit for the split-f16 conv kernel: parity tests, kernel timing, stage probe, bench + rocprofv3 kernel stats.
set -u
TAG=${1:-r01b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== conv tests"; timeout 600 python -m pytest tests/test_ops_gpu.py -q -k conv1d -x > $OUT/pytest_conv.log 2>&1; echo "exit $?"; tail -15 $OUT/pytest_conv.log
echo "== decoder tests"; timeout 600 python -m pytest tests/test_decoder_gpu.py tests/test_pipeline_gpu.py -q -x > $OUT/pytest_dec.log 2>&1; echo "exit $?"; tail -15 $OUT/pytest_dec.log
echo "== probe conv"; timeout 300 python tools/probe_conv.py > $OUT/probe_conv.log 2>&1; cat $OUT/probe_conv.log | tail -60
echo "== probe e2e"; timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e.log 2>&1; tail -6 $OUT/probe_e2e.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprof"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_prof.err ); echo "rocprof exit $?"
find /tmp/prof_$TAG -type f | head; for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/; done
head -25 $OUT/*kernel_stats.csv 2>/dev/null | cut -c1-200
