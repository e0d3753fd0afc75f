#!/bin/bash
# Round 2, visit c (first visit after the container was re-created): whole GPU suite on the C++ plans (ABI v9),
# smoke, long-form fault localisation, bench for every BASELINE config, rocprofv3 stats of the default bench.
set -u
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -40
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench (default = configs[1])"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-300 $OUT/bench_single.json
echo "== bench ST2_PLAN=python --single-stream"; ST2_PLAN=python timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single_pyplan.json 2> $OUT/bench_single_pyplan.err; cut -c1-300 $OUT/bench_single_pyplan.json
echo "== longform debug (graph=1 bucket=16)"; AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/debug_longform.py > $OUT/debug_longform.log 2>&1; echo "exit $?"; grep "dbg\|rror\|File" $OUT/debug_longform.log | tail -25
for c in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "exit $?"; cut -c1-400 $OUT/bench_$c.json; tail -3 $OUT/bench_$c.err
done
echo "== rocprof stats (--single-stream: un-overlapped per-kernel durations)"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o bench1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof1_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_single_kernel_stats.csv; done
head -16 $OUT/bench_single_kernel_stats.csv 2>/dev/null | cut -c1-220
echo "== probe lstm"; timeout 200 python tools/probe_lstm.py > $OUT/probe_lstm.log 2>&1; tail -8 $OUT/probe_lstm.log
