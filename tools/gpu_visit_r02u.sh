#!/bin/bash
# Round 2, visit u: MFMA issue order (operand reuse between neighbouring MFMAs) -- does it move the power-limited clock?
set -u
TAG=${1:-r02u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for b in xs_bench_k11_abl15 xs_bench_k11_ord1_abl15 xs_bench_k11_ord2_abl15 xs_bench_k11 xs_bench_k11_ord1_abl0 xs_bench_k11_ord2_abl0 xs_bench_k11_abl15 xs_bench_k11; do ./tools/bin/$b 11 1; done 2>&1 | tee $OUT/xs_bench_order.log
