#!/bin/bash
# Round 3, visit e: k = 3 conv tile / occupancy variants (tools/gemm_bench.hip -DK3_BENCH) on the shapes of the bench, the token
# GEMM rule in the library (default bench), kernel statistics of the default command.
set -u
TAG=${1:-r03e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== k3_bench"
for shape in "128 128 48001 32 10 1 0" "128 128 48001 32 10 0 0" "256 256 8000 32 10 1 0" "1024 1024 400 32 10 0 0" "512 512 800 32 10 1 0" "1024 1090 400 32 10 0 0"; do
  timeout 120 tools/bin/k3_bench $shape | tee -a $OUT/k3_bench.log
done
echo "== bench (default)"; timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<EOF
import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])
EOF
for s in single; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$s -o t -- python $R/bench.py --steps 3 --warmup 1 --calib-steps 0 --schedule $s --no-cpu-baseline > $R/$OUT/prof_$s.log 2>&1)
  find $OUT/prof_$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$s.csv
  f=$(find $OUT/prof_$s -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && gzip -c $f > $OUT/kernel_trace_$s.csv.gz
  rm -rf $OUT/prof_$s
done
head -30 $OUT/kernel_stats_single.csv | cut -c1-200
