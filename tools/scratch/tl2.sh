#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_GEMM_TABLE=gpurun_out/student_gemm_table.json
[ -f $FH_GEMM_TABLE ] || python tools/student_step.py --batch 128 > /dev/null 2>&1
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/student_step.py --batch 128 --steps 4 --warmup 2 > /dev/null 2>&1
python tools/timeline_list.py $(ls gpurun_out/tl/*/*kernel_trace.csv | head -1) ${1:-0} ${2:-100} > gpurun_out/timeline_list.txt
rm -rf gpurun_out/tl
wc -l gpurun_out/timeline_list.txt
