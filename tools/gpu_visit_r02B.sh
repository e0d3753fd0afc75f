#!/bin/bash
# Round 2, visit B: control run (default configuration on the same box) next to the LibriTTS / long-form configurations.
set -u
TAG=${1:-r02B}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in ljspeech libritts_istftnet longform ljspeech; do
  echo "== bench --config $c"; timeout 600 python bench.py --config $c --steps 5 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"
done
