#!/bin/bash
# A-B build of the fused conv WITHOUT the XCD-ordered column tiles: tools/bin/libst2_hip_f16s_dispatch.so = the library with the
# st2_conv1d_f16s_k*.hip instantiations compiled under -DST2_F16S_DISPATCH_ORDER (st2_conv1d_f16s_impl.h: tile = blockIdx.x).  Needs
# the regular library built first (reuses its other objects).  Compare with `python tools/probe_narrow.py [that library]`.
set -e
cd "$(dirname "$0")/.."
CS=styletts2_amd/csrc
mkdir -p tools/bin /tmp/f16s_disp
others=$(ls $CS/build/*.o | grep -v "st2_conv1d_f16s_k[0-9].o")
for k in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-slp-vectorize -Iinclude -I$CS -DST2_F16S_DISPATCH_ORDER \
    -c $CS/st2_conv1d_f16s_k$k.hip -o /tmp/f16s_disp/k$k.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libst2_hip_f16s_dispatch.so $others /tmp/f16s_disp/k0.o /tmp/f16s_disp/k1.o /tmp/f16s_disp/k2.o
echo built tools/bin/libst2_hip_f16s_dispatch.so
