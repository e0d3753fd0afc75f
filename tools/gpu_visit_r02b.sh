#!/bin/bash
# Round 2, visit b: whole GPU suite, same-box A/B of the conv kernels against the round-1 tree (ab_r01/, built from
# 49d11cc), long-form fault localisation, bench (two-stream and single).
set -u
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | head -40
echo "== A/B probe conv: round-1 tree"; ( cd ab_r01 && PROBE_KERNELS=f16s timeout 300 python tools/probe_conv.py > $R/$OUT/probe_conv_r01.log 2>&1 ); grep -o "'ks': [0-9]*, 'dil': [0-9]*\|'C': [0-9]*, 'L': [0-9]*\|'xs_plain': [0-9.]*\|'xs_res_stats': [0-9.]*\|'fused_pro3': [0-9.]*\|'act': [0-9.]*" $OUT/probe_conv_r01.log | paste - - - - - - | head -12
echo "== A/B probe conv: this tree"; PROBE_KERNELS=f16s timeout 300 python tools/probe_conv.py > $OUT/probe_conv.log 2>&1; grep -o "'ks': [0-9]*, 'dil': [0-9]*\|'C': [0-9]*, 'L': [0-9]*\|'xs_plain': [0-9.]*\|'xs_res_stats': [0-9.]*\|'fused_pro3': [0-9.]*\|'act': [0-9.]*" $OUT/probe_conv.log | paste - - - - - - | head -12
echo "== A/B probe conv: round-1 tree again (drift check)"; ( cd ab_r01 && PROBE_KERNELS=f16s timeout 300 python tools/probe_conv.py > $R/$OUT/probe_conv_r01_again.log 2>&1 ); grep -o "'xs_plain': [0-9.]*" $OUT/probe_conv_r01_again.log | paste - - - - - - - - - - 
echo "== longform debug (graph=1 bucket=16)"; AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/debug_longform.py > $OUT/debug_longform.log 2>&1; echo "exit $?"; grep "dbg\|rror\|File" $OUT/debug_longform.log | tail -25
echo "== longform debug (graph=0 bucket=16)"; DBG_GRAPH=0 AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/debug_longform.py > $OUT/debug_longform_nograph.log 2>&1; echo "exit $?"; grep "dbg\|rror\|File" $OUT/debug_longform_nograph.log | tail -12
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-330 $OUT/bench.json; tail -2 $OUT/bench.err
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-330 $OUT/bench_single.json
echo "== bench round-1 tree (same box)"; ( cd ab_r01 && timeout 600 python bench.py --no-cpu-baseline > $R/$OUT/bench_r01.json 2> $R/$OUT/bench_r01.err ); cut -c1-330 $OUT/bench_r01.json
echo "== bench --config longform"; timeout 600 python bench.py --config longform --steps 5 --no-cpu-baseline > $OUT/bench_longform.json 2> $OUT/bench_longform.err; echo "exit $?"; cut -c1-400 $OUT/bench_longform.json; tail -3 $OUT/bench_longform.err
