#!/bin/bash
# Round 2, visit k: per-workgroup timelines (s_memtime at start / k-loop end / epilogue end, HW_ID, XCC_ID).
set -u
TAG=${1:-r02k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
./tools/bin/xs_bench_k11_abl64 11 1 128 48001 32 1 1 3 0 $OUT/timeline_full.txt
./tools/bin/xs_bench_k11_abl75 11 1 128 48001 32 1 1 3 0 $OUT/timeline_epi_only.txt
gzip -f $OUT/timeline_full.txt $OUT/timeline_epi_only.txt
