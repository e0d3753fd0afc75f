#!/bin/bash
# Round 3, visit b: the Toom-Cook F(3,3) / F(4,4) / F(6,6) conv experiment (tools/wino_bench.hip) -- correctness
# against the host fp64 conv first, then timing next to the production kernel on the same box.
#   bash tools/build_xs_bench.sh 0 && gpurun --timeout 600 -- 'bash tools/gpu_visit_r03b.sh r03b'
set -u
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== wino check"; timeout 300 tools/bin/wino_bench check > $OUT/wino_check.log 2>&1; echo rc=$?; cat $OUT/wino_check.log
for k in 11 7; do
  echo "== production kernel k=$k (C=128, L=48001, B=32)"; timeout 120 tools/bin/xs_bench_0 $k 1 128 48001 32 1 1 10 | tee -a $OUT/xs_bench.log
  for tn in 2 1; do
    echo "== F(3,3) k=$k TN=$tn"; timeout 120 tools/bin/wino_bench $k 128 48001 32 10 $tn | tee -a $OUT/wino_bench.log
  done
  for occ in 2 3; do
    for wm in 4 2; do   # wm = 2: waves pair up on the same output rows (weight fragments shared through the vector L1)
      echo "== F(4,4) k=$k occ=$occ WM=$wm"; timeout 120 tools/bin/wino_bench $k 128 48001 32 10 1 1 1 44 $occ $wm | tee -a $OUT/wino_bench.log
    done
  done
  for wm in 4 2; do
    echo "== F(6,6) k=$k WM=$wm"; timeout 120 tools/bin/wino_bench $k 128 48001 32 10 1 1 1 66 2 $wm | tee -a $OUT/wino_bench.log
  done
  echo "== F(3,3) k=$k TN=1 WM=2"; timeout 120 tools/bin/wino_bench $k 128 48001 32 10 1 1 1 33 3 2 | tee -a $OUT/wino_bench.log
done
for d in 3 5; do   # dilated convs1 (residue-major output) and the convs2 behind them (residue-major input)
  echo "== production kernel k=11 dil=$d"; timeout 120 tools/bin/xs_bench_0 11 $d 128 48001 32 0 1 10 | tee -a $OUT/xs_bench.log
  echo "== F(3,3) k=11 dil=$d"; timeout 120 tools/bin/wino_bench 11 128 48001 32 10 2 $d 1 | tee -a $OUT/wino_bench.log
  echo "== F(3,3) k=11 dil=1 behind dil=$d"; timeout 120 tools/bin/wino_bench 11 128 48001 32 10 2 1 $d | tee -a $OUT/wino_bench.log
  echo "== F(4,4) k=11 dil=$d"; timeout 120 tools/bin/wino_bench 11 128 48001 32 10 1 $d 1 44 2 | tee -a $OUT/wino_bench.log
done
echo "== C=256, L=8000"; timeout 120 tools/bin/xs_bench_0 11 1 256 8000 32 1 1 10 | tee -a $OUT/xs_bench.log; timeout 120 tools/bin/wino_bench 11 256 8000 32 10 2 | tee -a $OUT/wino_bench.log; timeout 120 tools/bin/wino_bench 11 256 8000 32 10 1 1 1 44 2 | tee -a $OUT/wino_bench.log
