#!/bin/bash
# Round 2, visit o: weight prefetch distance 2 / 3 / 4 / 5 k-steps.
set -u
TAG=${1:-r02o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for n in 3 4 5 6; do ./tools/bin/xs_bench_k11_nset$n 11 1; ./tools/bin/xs_bench_k11_nset$n 11 1 256 8000 32 1 1; done 2>&1 | tee $OUT/xs_bench_nset.log
./tools/bin/xs_bench_k3_nset4 3 1 2>&1 | tee -a $OUT/xs_bench_nset.log
./tools/bin/xs_bench_k11_nset5_tl 11 1 128 48001 32 1 1 3 0 $OUT/tl_nset5.txt | tee -a $OUT/xs_bench_nset.log
gzip -f $OUT/tl_nset5.txt
