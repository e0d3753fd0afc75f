#!/bin/bash
# A/B visit for the multi-stream MRF: decoder + pipeline tests, bench in the four stream modes.
set -u
TAG=${1:-r01o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== decoder + pipeline tests"; timeout 900 python -m pytest tests/test_decoder_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/pytest_sel.log 2>&1; echo "exit $?"; tail -3 $OUT/pytest_sel.log
echo "== bench (default)"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-230 $OUT/bench.json; tail -2 $OUT/bench.err
echo "== bench ST2_MRF_STREAMS=0"; ST2_MRF_STREAMS=0 timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_mrf0.json 2> $OUT/bench_mrf0.err; cut -c1-230 $OUT/bench_mrf0.json
echo "== bench --single-stream"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; cut -c1-230 $OUT/bench_single.json
echo "== bench --single-stream ST2_MRF_STREAMS=0"; ST2_MRF_STREAMS=0 timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single_mrf0.json 2> $OUT/bench_single_mrf0.err; cut -c1-230 $OUT/bench_single_mrf0.json
echo "== probe e2e"; timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e.log 2>&1; tail -3 $OUT/probe_e2e.log
