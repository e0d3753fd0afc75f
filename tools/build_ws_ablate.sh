#!/bin/bash
# Measurement builds of the warp-specialised fused conv: tools/bin/libst2_hip_ws_abl<N>.so = the library with
# st2_conv1d_f16s_ws.h compiled under -DST2_WS_ABLATE=N (one stage of the kernel removed; results meaningless, times tell
# which stage bounds the launch).  Needs the regular library built first (reuses its other objects).
set -e
cd "$(dirname "$0")/.."
CS=styletts2_amd/csrc
mkdir -p tools/bin /tmp/ws_abl
others=$(ls $CS/build/*.o | grep -v "st2_conv1d_f16s_w[0-9].o")
EXTRA=${WS_EXTRA:-}
SUF=${WS_SUFFIX:-}
for n in "$@"; do
  for w in 0 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -DST2_WS_ABLATE=$n $EXTRA -Rpass-analysis=kernel-resource-usage -c $CS/st2_conv1d_f16s_w$w.hip -o /tmp/ws_abl/w${w}_$n.o 2> /tmp/ws_abl/w${w}_$n.log &
  done
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libst2_hip_ws_abl$n$SUF.so $others /tmp/ws_abl/w0_$n.o /tmp/ws_abl/w1_$n.o /tmp/ws_abl/w2_$n.o
  echo built tools/bin/libst2_hip_ws_abl$n$SUF.so
done
