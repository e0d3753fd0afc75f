#!/bin/bash
# Round 2, visit H: front-stream priority -1 (high) vs 0 on whatever box this lands on (slow boxes lose the overlap).
set -u
TAG=${1:-r02H}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for p in -1 0; do
  echo "== bench --front-priority $p"; timeout 200 python bench.py --no-cpu-baseline --front-priority $p > $OUT/bench_prio$p.json 2> $OUT/bench_prio$p.err; python -c "import json;r=json.load(open('$OUT/bench_prio$p.json'));print(r['ms_per_step'], r['value'], [(c['ks'],c['C_in'],c['L'],round(c['avg_launch_ms'],3)) for c in r['roofline']['classes'][:3]])"
done
