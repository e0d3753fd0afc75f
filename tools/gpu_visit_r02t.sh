#!/bin/bash
# Round 2, visit t: validation after the single generic epilogue body (static_for): conv / decoder / engine tests, bench
# (default, single, HiFi-GAN), rocprofv3 stats of the HiFi-GAN and long-form configurations.
set -u
TAG=${1:-r02t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest ops/decoder/engine"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decoder_gpu.py tests/test_engine_gpu.py tests/test_c_host.py -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['traffic'])"
echo "== bench single"; timeout 900 python bench.py --no-cpu-baseline --single-stream > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "import json;r=json.load(open('$OUT/bench_single.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'])"
echo "== rocprof stats hifigan"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2_$TAG -o bench2 -- python $R/bench.py --config libritts_hifigan --steps 3 --warmup 1 --no-cpu-baseline --single-stream > $R/$OUT/bench_prof_hifigan.json 2> $R/$OUT/bench_prof_hifigan.err ); echo "rocprof exit $?"
for f in $(find /tmp/prof2_$TAG -name '*kernel_stats.csv'); do cp $f $OUT/bench_hifigan_kernel_stats.csv; done
head -16 $OUT/bench_hifigan_kernel_stats.csv 2>/dev/null | cut -c1-150
python -c "import json;r=json.load(open('$OUT/bench_prof_hifigan.json'));print(r['ms_per_step'], r['value'])"
