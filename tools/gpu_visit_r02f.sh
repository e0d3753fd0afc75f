#!/bin/bash
# Round 2, visit f: ablation micro-benchmark of the dominant conv kernel (what does each part cost?), bench after the
# blocking host -> device copies were removed from the step.
set -u
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1 2 4 8 15; do for a in "11 1" "11 5"; do ./tools/bin/xs_bench_$m $a; done; done 2>&1 | tee $OUT/xs_bench_k11.log
for m in 0 4 15; do ./tools/bin/xs_bench_$m 7 3; ./tools/bin/xs_bench_$m 3 1; done 2>&1 | tee $OUT/xs_bench_k7_k3.log
./tools/bin/xs_bench_0 11 1 128 48001 32 0 0 2>&1 | tee -a $OUT/xs_bench_k11.log
./tools/bin/xs_bench_0 11 1 128 48001 8 1 1 2>&1 | tee -a $OUT/xs_bench_k11.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'])"
echo "== bench single"; timeout 600 python bench.py --single-stream --no-cpu-baseline > $OUT/bench_single.json 2> $OUT/bench_single.err; python -c "import json;r=json.load(open('$OUT/bench_single.json'));print(r['ms_per_step'], r['value'], r['config']['host_issue_ms_per_step'], r['roofline']['frac'])"
echo "== bench longform"; timeout 600 python bench.py --config longform --steps 5 --no-cpu-baseline > $OUT/bench_longform.json 2> $OUT/bench_longform.err; python -c "import json;r=json.load(open('$OUT/bench_longform.json'));print(r['ms_per_step'], r['value'], r['config']['first_chunk_latency_ms'])"; tail -2 $OUT/bench_longform.err
