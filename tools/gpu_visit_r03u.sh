#!/bin/bash
# Visit r03u: hardware counters of the fused conv (one-role and warp-specialised builds) on three vocoder layer shapes.
R=$(pwd); OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export PROBE_WS_PMC=1
(rocprofv3-avail list 2>/dev/null || rocprofv3 -L 2>/dev/null) > $OUT/r03u_counters_avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); echo "== pmc pass $i: $set"
  ( cd /tmp && timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_u_$i -o pmc -- python $R/tools/probe_ws.py > $R/$OUT/r03u_pmc_$i.log 2>&1 ); echo "pmc exit $?"
  python tools/pmc_summary.py /tmp/pmc_u_$i 2>&1 | grep "conv1d_f16s" > $OUT/r03u_pmc_pass$i.txt
  wc -l $OUT/r03u_pmc_pass$i.txt; tail -3 $OUT/r03u_pmc_$i.log
done
