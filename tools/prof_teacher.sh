#!/bin/bash
# rocprofv3 kernel stats of the teacher-only forward (B = FH_BENCH_BATCH, default 256, every kernel native): an un-profiled run first writes the per-shape GEMM
# table, the profiled run reuses it (FH_GEMM_TABLE) -- so the trace holds the forward's launches only, no tuner candidates
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_GEMM_TABLE=gpurun_out/teacher_gemm_table.json
rm -f $FH_GEMM_TABLE
python tools/teacher_step.py --native 1 --batch ${FH_BENCH_BATCH:-256} > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_teacher -- python tools/teacher_step.py --native 1 --batch ${FH_BENCH_BATCH:-256} --steps 8 --warmup 2 > gpurun_out/teacher_prof.log 2>&1
f=$(ls gpurun_out/prof_teacher/*/*kernel_stats.csv | head -1); cp $f gpurun_out/teacher_kernel_stats.csv; rm -rf gpurun_out/prof_teacher
grep "teacher fwd" gpurun_out/teacher_prof.log
