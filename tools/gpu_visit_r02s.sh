#!/bin/bash
# Round 2, visit s: graphed-front test, long-form with 128-column tiles at B = 1, default bench, HiFi-GAN kernel stats.
set -u
TAG=${1:-r02s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest pipeline"; timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_fullsize_gpu.py -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
echo "== bench longform"; timeout 600 python bench.py --config longform --steps 10 --no-cpu-baseline > $OUT/bench_longform.json 2> $OUT/bench_longform.err; python -c "import json;r=json.load(open('$OUT/bench_longform.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"; tail -2 $OUT/bench_longform.err
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['traffic'])"
echo "== rocprof stats hifigan"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$TAG -o bench2 -- python $R/bench.py --config libritts_hifigan --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_hifigan.json 2> $R/$OUT/bench_prof_hifigan.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof2_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_hifigan_kernel_stats.csv; done
head -14 $OUT/bench_hifigan_kernel_stats.csv 2>/dev/null | cut -c1-150
echo "== rocprof stats longform"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3_$TAG -o bench3 -- python $R/bench.py --config longform --steps 5 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_longform.json 2> $R/$OUT/bench_prof_longform.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof3_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_longform_kernel_stats.csv; done
head -14 $OUT/bench_longform_kernel_stats.csv 2>/dev/null | cut -c1-150
