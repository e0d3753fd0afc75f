#!/bin/bash
# Round 2, visit m: wave priorities (k loop vs epilogue).
set -u
TAG=${1:-r02m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for m in 0 128 384; do ./tools/bin/xs_bench_k11_abl$m 11 1; ./tools/bin/xs_bench_k11_abl$m 11 1 128 48001 32 0 1; done 2>&1 | tee $OUT/xs_bench_prio.log
./tools/bin/xs_bench_k11_abl192 11 1 128 48001 32 1 1 3 0 $OUT/timeline_noprio.txt 2>&1 | tee -a $OUT/xs_bench_prio.log
./tools/bin/xs_bench_k11_abl448 11 1 128 48001 32 1 1 3 0 $OUT/timeline_epiprio.txt 2>&1 | tee -a $OUT/xs_bench_prio.log
gzip -f $OUT/timeline_noprio.txt $OUT/timeline_epiprio.txt
