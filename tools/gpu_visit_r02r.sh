#!/bin/bash
# Round 2, visit r: C host on the module-level ABI, hipGraph-captured front (long-form), PMC passes over the dominant
# launch class (128-byte aligned row pitch, 32 x 256 wave tiles).
set -u
TAG=${1:-r02r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest c_host + graphed front"; timeout 900 python -m pytest tests/test_c_host.py tests/test_pipeline_gpu.py -m gpu -q --maxfail=10 -k "c_host or graphed_front or long_form" > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
echo "== bench longform (graphed front)"; timeout 600 python bench.py --config longform --steps 10 --no-cpu-baseline > $OUT/bench_longform.json 2> $OUT/bench_longform.err; python -c "import json;r=json.load(open('$OUT/bench_longform.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"; tail -2 $OUT/bench_longform.err
echo "== bench longform --eager-front"; timeout 600 python bench.py --config longform --steps 10 --no-cpu-baseline --eager-front > $OUT/bench_longform_eager.json 2> $OUT/bench_longform_eager.err; python -c "import json;r=json.load(open('$OUT/bench_longform_eager.json'));print(r['ms_per_step'], r['value'], r['config'].get('first_chunk_latency_ms'))"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); echo "== pmc pass $i: $set"
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- python $R/tools/probe_dom.py > $R/$OUT/pmc_$i.log 2>&1 ); echo "pmc exit $?"
  python tools/pmc_summary.py /tmp/pmc_${TAG}_$i > $OUT/pmc_pass$i.txt 2>&1; grep "conv1d_xs\|act_split\|instnorm\|copyBuffer" $OUT/pmc_pass$i.txt | cut -c1-60,100-200 | head -20
  grep probe_dom $OUT/pmc_$i.log
done
python tools/pmc_summary.py --json $OUT/pmc_dominant.json --kernel "conv1d_xs_kernel" /tmp/pmc_${TAG}_1 /tmp/pmc_${TAG}_2 | tail -1
