#!/bin/bash
# Round 2, visit v: split-K for skinny f16s convs (ABI v11): conv / engine / pipeline tests, bench for the configurations
# whose denoiser runs [B, F, N] k = 1 convs over ~100 tokens (long-form at B = 1, LibriTTS configs), default bench.
set -u
TAG=${1:-r02v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_sampler_gpu.py tests/test_c_host.py tests/test_pipeline_gpu.py -m gpu -q --maxfail=10 > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
for c in longform libritts_hifigan libritts_istftnet; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"; tail -1 $OUT/bench_$c.err
done
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'])"
