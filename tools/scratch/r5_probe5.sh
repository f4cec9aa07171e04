#!/bin/bash
# per-kernel VALU / MFMA / wait shares of the student step and the teacher forward (rocprofv3 --pmc, SQ counters only)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p5; mkdir -p $O
T=$O/table.json; TT=$O/ttable.json; rm -f $T $TT
FH_GEMM_TABLE=$T python tools/student_step.py --batch 256 --steps 3 > /dev/null 2>&1
FH_GEMM_TABLE=$TT python tools/teacher_step.py --native 1 --batch 256 --steps 2 > /dev/null 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf $O/p; FH_GEMM_TABLE=$T timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/p -- python tools/student_step.py --batch 256 --steps 2 --warmup 1 > $O/log_s$i.txt 2>&1
  python tools/pmc_sum.py $O/p > $O/student_pmc$i.txt 2>&1
  rm -rf $O/p; FH_GEMM_TABLE=$TT timeout 300 rocprofv3 --pmc $P --output-format csv -d $O/p -- python tools/teacher_step.py --native 1 --batch 256 --steps 2 --warmup 1 > $O/log_t$i.txt 2>&1
  python tools/pmc_sum.py $O/p > $O/teacher_pmc$i.txt 2>&1
done
rm -rf $O/p; wc -l $O/*.txt
