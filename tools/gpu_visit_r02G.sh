#!/bin/bash
# Round 2, visit G (last): one bench line per BASELINE configuration from the final library, same box.
set -u
TAG=${1:-r02G}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench"; timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'])"
for c in libritts_hifigan libritts_istftnet longform; do
  echo "== bench --config $c"; timeout 200 python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python -c "import json;r=json.load(open('$OUT/bench_$c.json'));print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['config'].get('first_chunk_latency_ms'))"
done
