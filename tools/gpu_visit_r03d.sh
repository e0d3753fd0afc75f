#!/bin/bash
# Round 3, box-class probe (cheap: ~70 s on a fast box).  Runs the default bench (all schedules calibrated) and decides from
# its own calibration whether this is one of the boxes on which two queues do not overlap (two-stream >= 0.95 x single).
# On such a box -- and always with FORCE=1 -- it collects what DESIGN.md section 6 could only infer so far: kernel traces with
# timestamps of both schedules (tools/trace_overlap.py), the same with the cooperative BiLSTM replaced by the single-CU
# kernel, an eager (un-graphed) front, equal stream priorities, and the fingerprint of the box.
#   gpurun --timeout 900 -- 'bash tools/gpu_visit_r03d.sh r03d'
set -u
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ uname -r; cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -i "fw_version\|num_xcc\|max_engine_clk_f\|num_cp_queues"; rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -i "power\|sclk\|mclk\|level"; rocminfo 2>&1 | grep -i "uuid" | tail -1; } > $OUT/box.txt 2>&1
echo "== bench (default)"; timeout 400 python bench.py --no-cpu-baseline --calib-partitioned > $OUT/bench.json 2> $OUT/bench.err
SLOW=$(python - <<EOF
import json;r=json.load(open('$OUT/bench.json'));c=r['config']['schedules_ms_per_step']
print(r['ms_per_step'], r['value'], r['config']['schedule'], c, file=__import__('sys').stderr)
print(1 if c['two-stream'] >= 0.95 * c['single'] else 0)
EOF
)
echo "slow-box class: $SLOW"
if [ "$SLOW" = "1" ] || [ "${FORCE:-0}" = "1" ]; then
  for s in two-stream single; do
    echo "== kernel trace, schedule $s"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$s -o t -- python $R/bench.py --steps 3 --warmup 1 --calib-steps 0 --schedule $s --no-cpu-baseline > $R/$OUT/prof_$s.log 2>&1)
    f=$(find $OUT/prof_$s -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/trace_overlap.py $f > $OUT/overlap_$s.json && python -c "
import json;r=json.load(open('$OUT/overlap_$s.json'));print({k:r[k] for k in ('span_ms','gpu_busy_ms','two_or_more_queues_active_ms','overlap_share_of_busiest_queue')}); print(r['queues']); print({k:v for k,v in r['families'].items() if 'lstm' in k or '<11' in k})"
    find $OUT/prof_$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$s.csv
    [ -n "$f" ] && gzip -c $f > $OUT/kernel_trace_$s.csv.gz
    rm -rf $OUT/prof_$s
  done
  echo "== two-stream, single-CU BiLSTM (no spin-waiting groups)"; timeout 200 python bench.py --schedule two-stream --lstm single --no-cpu-baseline > $OUT/bench_lstm_single.json 2> $OUT/bench_lstm_single.err; python -c "import json;r=json.load(open('$OUT/bench_lstm_single.json'));print(r['ms_per_step'])"
  echo "== single-stream, single-CU BiLSTM"; timeout 200 python bench.py --schedule single --lstm single --no-cpu-baseline > $OUT/bench_lstm_single_1s.json 2> $OUT/bench_lstm_single_1s.err; python -c "import json;r=json.load(open('$OUT/bench_lstm_single_1s.json'));print(r['ms_per_step'])"
  echo "== two-stream, eager front"; timeout 200 python bench.py --schedule two-stream --eager-front --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err; python -c "import json;r=json.load(open('$OUT/bench_eager.json'));print(r['ms_per_step'])"
  echo "== two-stream, equal priorities"; timeout 200 python bench.py --schedule two-stream --front-priority 0 --no-cpu-baseline > $OUT/bench_prio0.json 2> $OUT/bench_prio0.err; python -c "import json;r=json.load(open('$OUT/bench_prio0.json'));print(r['ms_per_step'])"
  echo "== two-stream, GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 200 python bench.py --schedule two-stream --no-cpu-baseline > $OUT/bench_q2.json 2> $OUT/bench_q2.err; python -c "import json;r=json.load(open('$OUT/bench_q2.json'));print(r['ms_per_step'])"
fi
