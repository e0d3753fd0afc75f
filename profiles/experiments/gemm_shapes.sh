#!/bin/bash
# tools/bin/gemm_bench over the (M, K) shapes of the token GEMMs on the path (N = 3 200 merged tokens at B = 32):
# denoiser q 512x1024, kv / o-in 1024x1024, o 1024x512, f1 2048x1024, f2 1024x2048; PL-BERT qkv 2304x768, o 768x768,
# ffn 2048x768 / 768x2048; bert_encoder 512x768.
cd "$(dirname "$0")/.."
for mk in "1024 1024" "2048 1024" "1024 2048" "512 1024" "1024 512" "2304 768" "768 768" "2048 768" "768 2048" "512 768"; do
  tools/bin/gemm_bench $mk 3200 1 ${REPS:-30} ${RES:-0} 0 | grep -E "library|128x64  c64|stream-K|refused"
done
