#!/bin/bash
# Round 2, visit j: each memory stream of the conv kernel alone on top of the pure MFMA loop; zero-data runs (clock).
set -u
TAG=${1:-r02j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for m in 15 11 7 13 14 0; do ./tools/bin/xs_bench_k11_abl$m 11 1; done 2>&1 | tee $OUT/xs_bench_streams.log
for m in 15 0; do ./tools/bin/xs_bench_k11_abl$m 11 1 128 48001 32 1 1 10 1; done 2>&1 | tee -a $OUT/xs_bench_streams.log
