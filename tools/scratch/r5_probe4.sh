#!/bin/bash
# PMC passes over the layer-2 conv data gradient probe (tools/scratch/dgrad_probe.py), one configuration per run
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p4; mkdir -p $O; rm -f $O/pmc.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P3="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
P5="SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"
for cfg in 31999,1,0,4 31999,1,0,2 32000,0,1,8 31999,0,0,4; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1)); rm -rf $O/p
    CFG=$cfg rocprofv3 --pmc $P --output-format csv -d $O/p -- python tools/scratch/dgrad_probe.py > $O/log.txt 2>&1
    echo "cfg $cfg pass $i: $(python tools/pmc_sum.py $O/p 2>&1 | grep -i gemm | head -2)" >> $O/pmc.txt
  done
done
rm -rf $O/p; cat $O/pmc.txt
