#!/bin/bash
# Round 2, visit h: micro-benchmark of the transposed-accumulator epilogue (16-byte stores).
set -u
TAG=${1:-r02h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for a in "11 1" "11 5" "7 3" "3 1" "11 1 128 48001 32 0 1" "11 1 128 48001 32 0 0" "11 1 256 8000 32 1 1" "7 1 256 8000 32 1 1" "3 1 256 8000 32 1 1"; do ./tools/bin/xs_bench_0 $a; done 2>&1 | tee $OUT/xs_bench_epilogue16.log
