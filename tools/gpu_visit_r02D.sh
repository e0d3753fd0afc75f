#!/bin/bash
# Round 2, visit D: same-box A/B of the LibriTTS configurations: tree of visit z (per-utterance [B, F, N] denoiser layout) vs
# this tree (token-merged storage, q / kv per utterance, o / f1 / f2 merged).
set -u
TAG=${1:-r02D}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { python -c "import json;r=json.load(open('$1'));print(r['ms_per_step'], r['value'])"; }
for c in libritts_hifigan libritts_istftnet; do
  echo "== r02z tree $c"; ( cd ab_r02z && timeout 600 python bench.py --config $c --no-cpu-baseline > $R/$OUT/bench_r02z_tree_$c.json 2> $R/$OUT/bench_r02z_tree_$c.err ); run $OUT/bench_r02z_tree_$c.json
  echo "== this tree $c"; timeout 600 python bench.py --config $c --no-cpu-baseline > $OUT/bench_this_$c.json 2> $OUT/bench_this_$c.err; run $OUT/bench_this_$c.json
done
