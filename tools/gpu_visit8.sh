#!/bin/bash
# Visit for the tiled ConvTranspose interleave: ops + decoder tests, bench in both stream modes, single-stream stats.
set -u
TAG=${1:-r01u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decoder_gpu.py -m gpu -x -q -k "transpose or decoder or harmonic or end_to_end" > $OUT/pytest_sel.log 2>&1; echo "exit $?"; tail -4 $OUT/pytest_sel.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-230 $OUT/bench.json
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-230 $OUT/bench_single.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o bench1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof1_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_single_kernel_stats.csv; done
grep "convt_interleave\|instnorm_stats" $OUT/bench_single_kernel_stats.csv | cut -c1-200
