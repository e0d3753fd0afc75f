#!/bin/bash
# Visit r03s: which stage bounds the warp-specialised fused conv -- the probe on the library and on the six measurement builds.
out=gpurun_out; mkdir -p $out
timeout 200 python tools/probe_ws.py > $out/r03s_probe_ws.log 2>&1
for n in 1 2 3 4 5 6; do
  echo "=== ST2_WS_ABLATE=$n" >> $out/r03s_probe_ws_abl.log
  timeout 100 python tools/probe_ws.py tools/bin/libst2_hip_ws_abl$n.so >> $out/r03s_probe_ws_abl.log 2>&1
done
grep -v "amdgpu.ids" $out/r03s_probe_ws_abl.log
