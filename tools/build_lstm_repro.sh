#!/bin/bash
# Builds tools/lstm_load_repro.hip in three flavours of the conv kernel (default / no s_setprio / predicated staging) into
# gpurun-visible binaries tools/bin/lstm_load_repro{,_noprio,_pred}.
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/bin
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Iinclude -Istyletts2_amd/csrc -DST2_XS_NARROW_ALL=1 tools/lstm_load_repro.hip -ldl"
( $B -o tools/bin/lstm_load_repro ) &
( $B -DST2_XS_SETPRIO=0 -o tools/bin/lstm_load_repro_noprio ) &
( $B -DST2_XS_PRED_STAGE=1 -o tools/bin/lstm_load_repro_pred ) &
wait
ls -la tools/bin/lstm_load_repro*
