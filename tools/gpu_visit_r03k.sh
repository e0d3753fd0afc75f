#!/bin/bash
# Round 3, visit k: attention / ConvTranspose-interleave staging with batched loads, iSTFT polar -> cartesian once per (frame,
# bin): GPU suite, default bench, single-stream kernel statistics.
set -u
TAG=${1:-r03k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu"; timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
echo "== bench"; timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['schedule'], r['config']['schedules_ms_per_step'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof1 -o t -- python $R/bench.py --steps 5 --warmup 1 --calib-steps 0 --schedule single --no-cpu-baseline > $R/$OUT/bench_prof_single.json 2> $R/$OUT/bench_prof_single.err)
find $OUT/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_single_kernel_stats.csv; rm -rf $OUT/prof1
grep "attention\|interleave\|istft\|style_fc" $OUT/bench_single_kernel_stats.csv | cut -c1-60,150-260
