#!/bin/bash
# Round 3, visit l: what each stream of a token GEMM costs (gemm_bench with the conv kernel's ablation switches: 4 = no epilogue,
# 2 = weight fragments loaded once, 8 = activations staged once, 15 = bare MFMA loop).
set -u
TAG=${1:-r03l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for shape in "2048 1024" "1024 1024" "1024 2048" "512 1024"; do
  for b in gemm_bench gemm_bench_abl4 gemm_bench_abl2 gemm_bench_abl8 gemm_bench_abl15; do
    echo "-- $b"; timeout 120 tools/bin/$b $shape 3200 1 20 | grep "library\|128x64  c64 occ3" | tee -a $OUT/gemm_ablate.log
  done
done
