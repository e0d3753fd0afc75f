#!/bin/bash
# One GPU-box visit: parity tests, smoke, stage probe, bench line, rocprofv3 kernel stats of the same bench command,
# then PMC passes (separate runs, --pmc only) over the dominant-launch probe.
# Usage (from the repo root on the GPU box): bash tools/gpu_visit.sh [tag] [skip-list]
set -u
TAG=${1:-r01c}
SKIP=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ ",$SKIP," == *",$1,"* ]]; }
if ! has pytest; then echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -12 $OUT/pytest_gpu.log; fi
if ! has smoke; then echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log; fi
if ! has probe; then echo "== probe e2e"; timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e.log 2>&1; tail -6 $OUT/probe_e2e.log
  echo "== probe lstm"; timeout 300 python tools/probe_lstm.py > $OUT/probe_lstm.log 2>&1; tail -8 $OUT/probe_lstm.log
  echo "== probe conv"; PROBE_KERNELS=f16s timeout 300 python tools/probe_conv.py > $OUT/probe_conv.log 2>&1; tail -40 $OUT/probe_conv.log; fi
if ! has bench; then echo "== bench (default: two streams, high-priority front)"; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-400 $OUT/bench.json; tail -4 $OUT/bench.err
  echo "== bench --front-priority 0"; timeout 600 python bench.py --front-priority 0 --no-cpu-baseline > $OUT/bench_prio0.json 2> $OUT/bench_prio0.err; cut -c1-230 $OUT/bench_prio0.json
  echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-230 $OUT/bench_single.json
  echo "== probe e2e libritts (HiFi-GAN, 10 steps)"; PROBE_TAG=libritts PROBE_STEPS=10 timeout 600 python tools/probe_e2e.py > $OUT/probe_e2e_libritts.log 2>&1; tail -3 $OUT/probe_e2e_libritts.log; fi
if ! has rocprof; then echo "== rocprof stats (default bench command)"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_prof.json 2> $R/$OUT/bench_prof.err ); echo "rocprof exit $?"
  for f in $(find /tmp/prof_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/; done
  echo "== rocprof stats (--single-stream: un-overlapped per-kernel durations)"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$TAG -o bench1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err ); echo "rocprof exit $?"
  for f in $(find /tmp/prof1_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_single_kernel_stats.csv; done
  head -12 $OUT/bench_single_kernel_stats.csv 2>/dev/null | cut -c1-180; fi
if ! has pmc; then
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    i=$((i+1)); echo "== pmc pass $i: $set"
    ( cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/tools/probe_dom.py > $R/$OUT/pmc_$i.log 2>&1 ); echo "pmc exit $?"
    python tools/pmc_summary.py /tmp/pmc_${TAG}_$i > $OUT/pmc_$i.txt 2>&1; grep -v "at::\|elementwise" $OUT/pmc_$i.txt | cut -c1-60,100-200 | head -40
    grep probe_dom $OUT/pmc_$i.log
  done
  python tools/pmc_summary.py --json $OUT/pmc_dominant.json --kernel "conv1d_xs_kernel" /tmp/pmc_${TAG}_1 /tmp/pmc_${TAG}_2 | tail -1
fi
