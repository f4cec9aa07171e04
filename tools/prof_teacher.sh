#!/bin/bash
# rocprofv3 kernel stats of the teacher-only forward (B=128, native frontend / attention / LayerNorm / fc1): the first forward's
# one-off GEMM tuning launches are part of the trace (the teacher tunes inside its first forward)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_teacher -- python tools/teacher_step.py --native 1 --batch 128 --steps 8 --warmup 2 > gpurun_out/teacher_prof.log 2>&1
f=$(ls gpurun_out/prof_teacher/*/*kernel_stats.csv | head -1); cp $f gpurun_out/teacher_kernel_stats.csv; rm -rf gpurun_out/prof_teacher
grep "teacher fwd" gpurun_out/teacher_prof.log
