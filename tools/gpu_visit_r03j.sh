#!/bin/bash
# Round 3, validation visit: whole GPU suite, smoke, the default bench line (with the reference CPU baseline from oracle/_ref),
# rocprofv3 kernel statistics of the default command, PMC passes over the dominant launch class, one bench line per config.
#   gpurun --timeout 2400 -- 'bash tools/gpu_visit_r03j.sh r03j'
set -u
TAG=${1:-r03j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ uname -r; cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -i "fw_version\|num_xcc\|max_engine_clk_f"; rocm-smi --showpower --showclocks --showperflevel 2>&1 | grep -i "power\|sclk\|mclk\|level"; } > $OUT/box.txt 2>&1
echo "== pytest -m gpu"; timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
echo "== bench (default, with cpu_baseline)"; timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<EOF
import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['kind'], r['cpu_baseline']['value'], r['cpu_baseline'].get('port',{}).get('value'))
EOF
echo "== rocprofv3 --kernel-trace --stats over the default command"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o t -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_prof.json 2> $R/$OUT/bench_prof.err); echo rc=$?
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_overlap.py $f --skip-ms 0 > $OUT/overlap_default.json; rm -rf $OUT/prof
head -12 $OUT/bench_kernel_stats.csv | cut -c1-150
echo "== rocprofv3 over --schedule single"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof1 -o t -- python $R/bench.py --steps 5 --warmup 1 --calib-steps 0 --schedule single --no-cpu-baseline > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err)
find $OUT/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_single_kernel_stats.csv; rm -rf $OUT/prof1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); echo "== pmc pass $i: $set"
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/tools/probe_dom.py > $R/$OUT/pmc_$i.log 2>&1 ); echo "pmc exit $?"
  python tools/pmc_summary.py /tmp/pmc_${TAG}_$i > $OUT/pmc_pass$i.txt 2>&1; grep "conv1d_xs\|act_split\|instnorm" $OUT/pmc_pass$i.txt | cut -c1-60,100-200 | head -8
done
python tools/pmc_summary.py --json $OUT/pmc_dominant.json --kernel "conv1d_xs_kernel" /tmp/pmc_${TAG}_1 /tmp/pmc_${TAG}_2 | tail -1
for cfg in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $cfg"; timeout 500 python bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python -c "import json;r=json.load(open('$OUT/bench_$cfg.json'));print(r['ms_per_step'], r['value'], r['config'].get('schedules_ms_per_step'), r['config'].get('first_chunk_latency_ms'), r['cpu_baseline']['kind'], r['cpu_baseline']['value'])"
done
