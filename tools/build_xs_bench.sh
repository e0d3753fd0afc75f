#!/bin/bash
# Builds tools/xs_bench.hip once per ablation mask into gpurun-visible binaries tools/bin/xs_bench_<mask>.
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/bin
MASKS=${*:-0 1 2 4 8 15 64}
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Iinclude \
      -Istyletts2_amd/csrc -DST2_XS_ABLATE=$m -DST2_XS_NARROW_ALL=1 tools/xs_bench.hip -o tools/bin/xs_bench_$m ) &
done
wait
ls -la tools/bin
