#!/bin/bash
# Round 2, visit g: phase-stagger experiment on the conv micro-benchmark.
set -u
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for m in 0 16 32 20 4; do ./tools/bin/xs_bench_$m 11 1; ./tools/bin/xs_bench_$m 7 3; ./tools/bin/xs_bench_$m 3 1; done 2>&1 | tee $OUT/xs_bench_stagger.log
for m in 0 16 32; do ./tools/bin/xs_bench_$m 11 1 128 48001 32 0 1; ./tools/bin/xs_bench_$m 11 1 256 8000 32 1 1; done 2>&1 | tee -a $OUT/xs_bench_stagger.log
