#!/bin/bash
# Graphed-sampler visit: sampler tests + smoke.
set -u
TAG=${1:-r01s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_sampler_gpu.py -m gpu -x -q > $OUT/pytest_sampler.log 2>&1; echo "exit $?"; tail -15 $OUT/pytest_sampler.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== latency probe B=1 (eager sampler)"; PROBE_B=1 timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e_b1.log 2>&1; tail -2 $OUT/probe_e2e_b1.log
echo "== latency probe B=1 (graphed sampler)"; PROBE_B=1 PROBE_GRAPH=1 timeout 300 python tools/probe_e2e.py > $OUT/probe_e2e_b1_graph.log 2>&1; tail -2 $OUT/probe_e2e_b1_graph.log
