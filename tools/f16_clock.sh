#!/bin/bash
# fp16 vs bf16 on identical launches: time (un-profiled), then cycles per launch from rocprofv3 --pmc (tools/f16_vs_bf16_clock.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f16clock; mkdir -p $O; rm -rf $O/p
python tools/f16_vs_bf16_clock.py 2>/dev/null | tee $O/time.txt
REPS=4 timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p -- python tools/f16_vs_bf16_clock.py > $O/pmc_run.txt 2>&1
python tools/pmc_sum.py $O/p | grep gemm | tee $O/pmc.txt
head -2 $(ls $O/p/*/*counter_collection.csv | head -1) > $O/csv_head.txt
rm -rf $O/p
