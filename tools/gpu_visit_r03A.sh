#!/bin/bash
# Visit r03A: the warp-specialised fused conv in the product -- its tests, the conv tests, then the HiFi-GAN configuration with a
# kernel profile (which launches make up configs[2] now).
R=$(pwd); OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "warp_specialised or conv1d" 2>&1 | tail -5 | tee $OUT/r03A_pytest_conv.log
timeout 300 python tools/probe_ws.py 2>&1 | grep -v amdgpu.ids > $OUT/r03A_probe_ws.log; tail -3 $OUT/r03A_probe_ws.log
timeout 400 python bench.py --config libritts_hifigan --no-cpu-baseline > $OUT/r03A_bench_libritts_hifigan.json 2> $OUT/r03A_bench_libritts_hifigan.err; tail -c 600 $OUT/r03A_bench_libritts_hifigan.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/profA -o t -- python $R/bench.py --config libritts_hifigan --steps 4 --warmup 1 --calib-steps 0 --schedule single --no-cpu-baseline > $R/$OUT/r03A_bench_prof_hifigan.json 2> $R/$OUT/r03A_bench_prof_hifigan.err)
find $OUT/profA -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03A_bench_hifigan_single_kernel_stats.csv; rm -rf $OUT/profA
head -25 $OUT/r03A_bench_hifigan_single_kernel_stats.csv | cut -c1-140
