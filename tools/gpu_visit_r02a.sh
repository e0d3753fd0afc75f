#!/bin/bash
# Round 2, visit a: full GPU test suite (ABI v9, new parity tests), smoke, BiLSTM hand-off probe, bench lines for all
# four BASELINE configs, rocprofv3 stats of the default bench command.
set -u
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -x --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== probe lstm"; timeout 300 python tools/probe_lstm.py > $OUT/probe_lstm.log 2>&1; tail -12 $OUT/probe_lstm.log
echo "== probe conv"; PROBE_KERNELS=f16s timeout 300 python tools/probe_conv.py > $OUT/probe_conv.log 2>&1; tail -30 $OUT/probe_conv.log
echo "== bench (default = configs[1])"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-600 $OUT/bench.json; tail -4 $OUT/bench.err
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-300 $OUT/bench_single.json
for c in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $c"; timeout 900 python bench.py --config $c --steps 5 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "exit $?"; cut -c1-500 $OUT/bench_$c.json; tail -3 $OUT/bench_$c.err
done
echo "== rocprof stats (--single-stream: un-overlapped per-kernel durations)"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o bench1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof1_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_single_kernel_stats.csv; done
head -14 $OUT/bench_single_kernel_stats.csv 2>/dev/null | cut -c1-200
