#!/bin/bash
# Round 2, visit n: shader clock under load (s_memtime vs the 100 MHz s_memrealtime), full kernel and pure MFMA loop, random and zero data.
set -u
TAG=${1:-r02n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
./tools/bin/xs_bench_k11_abl64 11 1 128 48001 32 1 1 3 0 $OUT/tl_full_random.txt
./tools/bin/xs_bench_k11_abl64 11 1 128 48001 32 1 1 3 1 $OUT/tl_full_zero.txt
./tools/bin/xs_bench_k11_abl79 11 1 128 48001 32 1 1 3 0 $OUT/tl_mfma_random.txt
./tools/bin/xs_bench_k11_abl79 11 1 128 48001 32 1 1 3 1 $OUT/tl_mfma_zero.txt
gzip -f $OUT/tl_*.txt
