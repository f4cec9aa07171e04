#!/bin/bash
# student step with the weight gradients serialised on the main stream: every kernel alone on the device -> true per-kernel times
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_WGRAD_STREAM=0 FH_GEMM_TABLE=gpurun_out/student_gemm_table_serial.json
rm -f $FH_GEMM_TABLE; mkdir -p gpurun_out/r5ser
timeout 600 python tools/student_step.py --batch 256 > gpurun_out/r5ser/plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ser -- python tools/student_step.py --batch 256 --steps 8 --warmup 2 > gpurun_out/r5ser/prof.log 2>&1
f=$(ls gpurun_out/prof_ser/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r5ser/kernel_stats.csv; rm -rf gpurun_out/prof_ser
grep "student step" gpurun_out/r5ser/plain.log gpurun_out/r5ser/prof.log
