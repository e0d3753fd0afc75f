#!/bin/bash
# Round 2, visit A: token-merged storage for the multispeaker denoiser (q / kv per utterance, o / f1 / f2 merged),
# duration-sum status bit: sampler / engine / ops tests, LibriTTS configurations.
set -u
TAG=${1:-r02A}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_engine_gpu.py tests/test_ops_gpu.py -m gpu -q --maxfail=10 -k "sampler or denoiser or expand or engine" > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
for c in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"
done
