#!/bin/bash
# Round 2, visit C: same-box A/B of the default bench: the tree of visit z (ab_r02z/, commit e84c032) vs this tree.
set -u
TAG=${1:-r02C}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { python -c "import json;r=json.load(open('$1'));print(r['ms_per_step'], r['value'], r['config'].get('host_issue_ms_per_step'), [ (c['ks'],c['C_in'],c['L'],round(c['avg_launch_ms'],3)) for c in r['roofline']['classes'][:4]])"; }
echo "== r02z tree"; ( cd ab_r02z && timeout 600 python bench.py --no-cpu-baseline > $R/$OUT/bench_r02z_tree.json 2> $R/$OUT/bench_r02z_tree.err ); run $OUT/bench_r02z_tree.json
echo "== this tree"; timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_this.json 2> $OUT/bench_this.err; run $OUT/bench_this.json
echo "== this tree --eager-front"; timeout 600 python bench.py --no-cpu-baseline --eager-front > $OUT/bench_this_eager.json 2> $OUT/bench_this_eager.err; run $OUT/bench_this_eager.json
echo "== r02z tree again"; ( cd ab_r02z && timeout 600 python bench.py --no-cpu-baseline > $R/$OUT/bench_r02z_tree2.json 2> $R/$OUT/bench_r02z_tree2.err ); run $OUT/bench_r02z_tree2.json
