#!/bin/bash
# Round 2, visit i: 32 x 256 wave-tile variant (micro-benchmark), correctness of the transposed-accumulator epilogue in
# the library (ops / decoder / engine / pipeline tests), bench with the roofline leg on the C timing hook.
set -u
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for a in "11 1" "11 5" "11 1 128 48001 32 0 1" "11 1 256 8000 32 1 1"; do ./tools/bin/xs_bench_tn8_k11 $a; done 2>&1 | tee $OUT/xs_bench_tn8.log
./tools/bin/xs_bench_tn8_k11_abl4 11 1 2>&1 | tee -a $OUT/xs_bench_tn8.log
./tools/bin/xs_bench_tn8_k7 7 3 2>&1 | tee -a $OUT/xs_bench_tn8.log
./tools/bin/xs_bench_tn8_k3 3 1 2>&1 | tee -a $OUT/xs_bench_tn8.log
echo "== pytest ops/decoder/engine/pipeline"; timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_decoder_gpu.py tests/test_engine_gpu.py tests/test_pipeline_gpu.py tests/test_style_gpu.py -m gpu -q --maxfail=20 > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_sel.log | head -30
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms']); [print(c) for c in r['roofline']['classes']]"; tail -2 $OUT/bench.err
echo "== bench single"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "import json;r=json.load(open('$OUT/bench_single.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms']); [print(c) for c in r['roofline']['classes']]"
