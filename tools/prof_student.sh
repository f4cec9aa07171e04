#!/bin/bash
# rocprofv3 kernel stats of the student-only step (B = FH_BENCH_BATCH, default 256): first an un-profiled run writes the GEMM table, then the profiled run reuses it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_GEMM_TABLE=gpurun_out/student_gemm_table.json
rm -f $FH_GEMM_TABLE
python tools/student_step.py --batch ${FH_BENCH_BATCH:-256} > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_student -- python tools/student_step.py --batch ${FH_BENCH_BATCH:-256} --steps 8 --warmup 2 > gpurun_out/student_prof.log 2>&1
f=$(ls gpurun_out/prof_student/*/*kernel_stats.csv | head -1); cp $f gpurun_out/student_kernel_stats.csv; rm -rf gpurun_out/prof_student
grep "student step" gpurun_out/student_prof.log
