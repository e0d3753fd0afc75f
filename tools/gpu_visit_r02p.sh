#!/bin/bash
# Round 2, visit p: k-loop duration of a workgroup that is alone / with one / with two others on its CU.
set -u
TAG=${1:-r02p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for b in 2 4 6; do
  ./tools/bin/xs_bench_k11_abl64 11 1 128 16384 $b 1 1 3 0 $OUT/tl_full_b$b.txt
  ./tools/bin/xs_bench_k11_abl79 11 1 128 16384 $b 1 1 3 0 $OUT/tl_mfma_b$b.txt
done 2>&1 | tee $OUT/xs_bench_occupancy.log
gzip -f $OUT/tl_*.txt
