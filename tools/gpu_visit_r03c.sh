#!/bin/bash
# Round 3, visit c: (1) token-GEMM tile shapes (tools/gemm_bench.hip) on the denoiser / PL-BERT shapes, (2) kernel traces with
# timestamps of the two-stream and single-stream schedules (the rocprofv3 runs of visit a died at exit with the CU-masked
# streams alive) -> tools/trace_overlap.py, (3) rocprofv3 over the default (auto) command again now that the masked streams
# are destroyed before exit.
#   gpurun --timeout 1200 -- 'bash tools/gpu_visit_r03c.sh r03c'
set -u
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -i "fw_version\|num_xcc\|max_engine_clk_f"; rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk"; } > $OUT/box.txt 2>&1
echo "== gemm_bench"
for shape in "2048 1024" "1024 2048" "1024 1024" "512 1024" "1024 512" "2304 768" "768 768" "2048 768" "768 2048" "512 768"; do
  timeout 120 tools/bin/gemm_bench $shape 3200 1 20 | tee -a $OUT/gemm_bench.log
done
for s in two-stream single; do
  echo "== kernel trace, schedule $s"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$s -o t -- python $R/bench.py --steps 3 --warmup 1 --calib-steps 0 --schedule $s --no-cpu-baseline > $R/$OUT/prof_$s.log 2>&1)
  tail -2 $OUT/prof_$s.log | cut -c1-300
  f=$(find $OUT/prof_$s -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py $f > $OUT/overlap_$s.json && python -c "
import json;r=json.load(open('$OUT/overlap_$s.json'));print({k:r[k] for k in ('span_ms','gpu_busy_ms','two_or_more_queues_active_ms','overlap_share_of_busiest_queue')}); print(r['queues']); print({k:v for k,v in r['families'].items() if 'lstm' in k or '<11' in k})"
  find $OUT/prof_$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$s.csv
  [ -n "$f" ] && gzip -c $f > $OUT/kernel_trace_$s.csv.gz
  rm -rf $OUT/prof_$s
done
echo "== rocprofv3 over the default command (auto schedule, masked streams closed at exit)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_auto -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$OUT/prof_auto.log 2>&1); echo rc=$?
tail -2 $OUT/prof_auto.log | cut -c1-400
find $OUT/prof_auto -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_auto.csv
rm -rf $OUT/prof_auto
echo "== bench (default)"; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<EOF
import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['kind'], r['cpu_baseline']['value'], r['cpu_baseline'].get('port'))
EOF
