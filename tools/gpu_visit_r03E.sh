#!/bin/bash
# Visit r03E: the lighter prologue (v_med3 clamp, channel tail through the parameter table, no SLP packing in the one-role fused
# kernel) -- op-level, decoder and full-size tests, then the default and the HiFi-GAN bench lines.
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_decoder_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/r03E_pytest.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/r03E_bench.json 2> $OUT/r03E_bench.err
timeout 300 python bench.py --config libritts_hifigan --no-cpu-baseline > $OUT/r03E_bench_libritts_hifigan.json 2> $OUT/r03E_bench_libritts_hifigan.err
python - <<PY
import json
for f in ("r03E_bench.json", "r03E_bench_libritts_hifigan.json"):
    d = json.load(open("$OUT/" + f)); print(f, d["ms_per_step"], d["value"], d["config"].get("schedules_ms_per_step"), d["roofline"]["frac"])
PY
