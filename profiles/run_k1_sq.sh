#!/bin/bash
# SQ / TCP counters of the product kernel K1 (VERDICT r04 "Next" #2): fp64 value stream and value dictionary, the
# one-workgroup-per-chunk kernel (mode 0) and the resident pipelined one (mode 1).  One counter set per pass, never combined
# with trace domains.  Run on the GPU box from the repo root:   bash profiles/run_k1_sq.sh && python profiles/summarize_k1_sq.py r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/k1_sq
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
MODES=${MODES:-"0"}
for vd in 0 1; do
  for mode in $MODES; do
    i=0
    for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
               "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
               "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_LEVEL_WAVES" \
               "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
               "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" \
               "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
      i=$((i+1))
      tag=vd${vd}_m$(echo $mode | tr ',' 'w')_s$i
      PA_K1_VD=$vd timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/$tag -o c -- python $R/tools/probe/k1_time.py 256 2 > $O/$tag.log 2>&1
      echo "$tag rc=$? $(tail -1 $O/$tag.log | cut -c1-160)"
    done
  done
done
