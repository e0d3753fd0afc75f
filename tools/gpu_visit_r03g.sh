#!/bin/bash
# Round 3, visit g: token-GEMM body with deep activation prefetch (gemm_bench "G" variants), the multispeaker denoiser's q / kv
# projections as token-merged GEMMs (st2_act_split gb_seg) in both LibriTTS configurations, whole GPU suite.
set -u
TAG=${1:-r03g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for shape in "1024 1024" "1024 2048" "2048 1024" "512 1024" "2304 768" "768 2048" "768 768"; do
  timeout 120 tools/bin/gemm_bench $shape 3200 1 20 | grep "library\|128x64  c64 occ3\|G " | tee -a $OUT/gemm_bench.log
done
for cfg in libritts_hifigan libritts_istftnet ljspeech; do
  echo "== bench $cfg"; timeout 400 python bench.py --config $cfg --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python -c "import json;r=json.load(open('$OUT/bench_$cfg.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])"
done
echo "== pytest -m gpu"; timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
