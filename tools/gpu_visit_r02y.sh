#!/bin/bash
# Round 2, visit y: fused kernel on the shared (transposed, 16-byte, statistics-emitting) epilogue: conv tests, then
# fused vs activation-pass + xs per layer shape at B = 32 (which layers should take the fused kernel?).
set -u
TAG=${1:-r02y}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest ops"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q --maxfail=10 -k "conv1d or split_f16 or activate" > $OUT/pytest_sel.log 2>&1; echo "exit $?" | tee -a $OUT/pytest_sel.log; grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_sel.log | head -20
echo "== probe conv B=32"; PROBE_B=32 timeout 600 python tools/probe_conv.py > $OUT/probe_conv_b32.log 2>&1; echo "exit $?"
grep -o "'C': [0-9]*, 'L': [0-9]*, 'ks': [0-9]*, 'dil': [0-9]*\|'fused_res_stats': [0-9.]*\|'act': [0-9.]*\|'xs_res_stats': [0-9.]*\|'layer_fused_ms': [0-9.]*\|'layer_xs_ms': [0-9.]*" $OUT/probe_conv_b32.log | paste - - - - - - | column -t
