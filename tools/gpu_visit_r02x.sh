#!/bin/bash
# Round 2, visit x: host issue time with the device-only front replayed from a hipGraph (bench --graph-front).
set -u
TAG=${1:-r02x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for mode in "" "--graph-front" "--single-stream" "--single-stream --graph-front"; do
  n=$(echo "bench$mode" | tr -d ' ' | tr '-' '_')
  echo "== bench $mode"; timeout 600 python bench.py --no-cpu-baseline $mode > $OUT/$n.json 2> $OUT/$n.err; python -c "import json;r=json.load(open('$OUT/$n.json'));print(r['ms_per_step'], r['value'], 'host issue', r['config']['host_issue_ms_per_step'], r['roofline']['frac'])"; tail -1 $OUT/$n.err
done
