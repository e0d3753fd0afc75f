#!/bin/bash
# Round 3, visit f: weight-prefetch depth of the token GEMMs (gemm_bench built with ST2_XS_NSET = 3 / 5 / 7), kernel statistics
# of the HiFi-GAN configuration (BASELINE.json configs[2]) and its bench line.
set -u
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "1024 1024" "1024 2048" "2048 1024" "512 1024" "2304 768" "768 2048"; do
  for b in gemm_bench gemm_bench_nset5 gemm_bench_nset7; do
    echo "-- $b"; timeout 120 tools/bin/$b $shape 3200 1 20 | grep "library\|128x64  c64\|128x128 c64" | tee -a $OUT/gemm_nset.log
  done
done
echo "== bench libritts_hifigan"; timeout 400 python bench.py --config libritts_hifigan --no-cpu-baseline > $OUT/bench_hifigan.json 2> $OUT/bench_hifigan.err
python - <<EOF
import json;r=json.load(open('$OUT/bench_hifigan.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])
EOF
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_h -o t -- python $R/bench.py --config libritts_hifigan --steps 3 --warmup 1 --calib-steps 0 --schedule single --no-cpu-baseline > $R/$OUT/prof_h.log 2>&1)
find $OUT/prof_h -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_hifigan_single.csv
rm -rf $OUT/prof_h
head -24 $OUT/kernel_stats_hifigan_single.csv | cut -c1-180
