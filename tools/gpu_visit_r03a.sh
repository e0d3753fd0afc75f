#!/bin/bash
# Round 3, visit a: (1) fingerprint of the box, (2) the three schedules of bench.py side by side (single / two-stream /
# CU-partitioned) with rocprofv3 kernel traces (timestamps) of each -> tools/trace_overlap.py answers whether the two queues
# run concurrently on THIS box and what the cooperative BiLSTM costs there, (3) C++ front vs Python front, (4) the
# Toom-Cook conv experiment (tools/gpu_visit_r03b.sh).
#   gpurun --timeout 1500 -- 'bash tools/gpu_visit_r03a.sh r03a'
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
  echo "== uname"; uname -a
  echo "== amdgpu"; cat /sys/module/amdgpu/version 2>/dev/null; ls /sys/class/kfd/kfd/topology/nodes/ 2>/dev/null
  echo "== env"; env | grep -i "^HSA\|^HIP\|^GPU_\|^ROC\|^AMD" 
  echo "== rocm-smi"; rocm-smi --showclocks --showpower --showperflevel --showmemuse --showuse 2>&1 | head -60
  echo "== rocm-smi partitions"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20
  echo "== rocminfo"; rocminfo 2>&1 | grep -i "name:\|compute unit\|max clock\|queue\|wavefront\|firmware\|uuid\|SDMA\|Features" | head -60
  echo "== kfd node props"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo $n; grep -i "simd_count\|cu_per\|array_count\|num_xcc\|fw_version\|max_engine_clk\|num_cp_queues\|num_sdma" $n/properties 2>/dev/null; done
  echo "== kfd params"; for p in sched_policy hws_max_conc_proc max_num_of_queues_per_device queue_preemption_timeout_ms halt_if_hws_hang; do echo -n "$p="; cat /sys/module/amdgpu/parameters/$p 2>/dev/null || echo "?"; done
} > $OUT/box.txt 2>&1
echo "== bench auto (all schedules calibrated)"; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<EOF
import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['kind'], r['cpu_baseline']['value'])
EOF
for cus in 16 64; do
  echo "== bench partitioned front-cus $cus"; timeout 200 python bench.py --schedule partitioned --front-cus $cus --no-cpu-baseline --calib-steps 0 > $OUT/bench_part$cus.json 2> $OUT/bench_part$cus.err
  python -c "import json;r=json.load(open('$OUT/bench_part$cus.json'));print(r['ms_per_step'], r['value'])"
done
for s in two-stream partitioned single; do
  echo "== kernel trace, schedule $s"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$s -o t -- python $R/bench.py --steps 3 --warmup 1 --calib-steps 0 --schedule $s --no-cpu-baseline > $R/$OUT/prof_$s.log 2>&1)
  f=$(find $OUT/prof_$s -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py $f > $OUT/overlap_$s.json && python -c "
import json;r=json.load(open('$OUT/overlap_$s.json'));print({k:r[k] for k in ('span_ms','gpu_busy_ms','two_or_more_queues_active_ms','overlap_share_of_busiest_queue')}); print(r['queues']); print({k:v for k,v in r['families'].items() if 'lstm' in k or '<11' in k})"
  find $OUT/prof_$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$s.csv
  [ -n "$f" ] && gzip -c $f > $OUT/kernel_trace_$s.csv.gz
  rm -rf $OUT/prof_$s
done
for f in engine; do   # C++ front (one C-ABI call) against the Python front above
  echo "== bench ST2_FRONT=$f (graphed)"; ST2_FRONT=$f timeout 200 python bench.py --no-cpu-baseline > $OUT/bench_graph_$f.json 2> $OUT/bench_graph_$f.err
  python -c "import json;r=json.load(open('$OUT/bench_graph_$f.json'));print(r['ms_per_step'], r['value'], r['config']['schedules_ms_per_step'], r['config']['host_issue_ms_per_step'])"
  echo "== bench --eager-front ST2_FRONT=$f"; ST2_FRONT=$f timeout 200 python bench.py --eager-front --no-cpu-baseline > $OUT/bench_eager_$f.json 2> $OUT/bench_eager_$f.err
  python -c "import json;r=json.load(open('$OUT/bench_eager_$f.json'));print(r['ms_per_step'], r['value'], r['config']['schedules_ms_per_step'], r['config']['host_issue_ms_per_step'])"
done
echo "== style plan probe"; timeout 200 python tools/probe_style_plan.py > $OUT/style_plan.log 2>&1; tail -4 $OUT/style_plan.log
echo "== wino"; bash tools/gpu_visit_r03b.sh $TAG 2>&1 | tail -80
