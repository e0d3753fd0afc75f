#!/bin/bash
# Round 2, visit l: epilogue with all residual loads in flight at once (micro-benchmark + timeline).
set -u
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for a in "11 1" "11 5" "11 1 128 48001 32 0 1" "11 1 256 8000 32 1 1"; do ./tools/bin/xs_bench_k11_abl0 $a; done 2>&1 | tee $OUT/xs_bench_epi_prefetch.log
./tools/bin/xs_bench_k7_abl0 7 3 2>&1 | tee -a $OUT/xs_bench_epi_prefetch.log
./tools/bin/xs_bench_k3_abl0 3 1 2>&1 | tee -a $OUT/xs_bench_epi_prefetch.log
./tools/bin/xs_bench_k11_abl64 11 1 128 48001 32 1 1 3 0 $OUT/timeline_full.txt 2>&1 | tee -a $OUT/xs_bench_epi_prefetch.log
gzip -f $OUT/timeline_full.txt
