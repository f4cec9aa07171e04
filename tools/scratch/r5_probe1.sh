#!/bin/bash
# round 5, first probe: per-launch GEMM traces of the student step and the teacher forward in bf16 and fp16 at B = 256
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p1; mkdir -p $O
for dt in bf16 float16; do
  T=$O/table_$dt.json; rm -f $T
  FH_GEMM_TABLE=$T python tools/student_step.py --batch 256 --steps 6 --dtype $dt > $O/student_$dt.tune.log 2>&1
  rm -f $O/trace_student_$dt.txt
  FH_GEMM_TABLE=$T FH_GEMM_TRACE=$O/trace_student_$dt.txt python tools/student_step.py --batch 256 --steps 8 --dtype $dt > $O/student_$dt.log 2>&1
  rm -f $O/trace_teacher_$dt.txt
  TT=$O/ttable_$dt.json; rm -f $TT
  FH_GEMM_TABLE=$TT python tools/teacher_step.py --native 1 --batch 256 --steps 3 --dtype $dt > $O/teacher_$dt.tune.log 2>&1
  FH_GEMM_TABLE=$TT FH_GEMM_TRACE=$O/trace_teacher_$dt.txt python tools/teacher_step.py --native 1 --batch 256 --steps 8 --dtype $dt > $O/teacher_$dt.log 2>&1
  tail -2 $O/student_$dt.log $O/teacher_$dt.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16 -- python tools/student_step.py --batch 256 --steps 6 --dtype float16 > $O/prof_f16.log 2>&1
cp $(ls $O/prof_f16/*/*kernel_stats.csv | head -1) $O/student_f16_kernel_stats.csv; rm -rf $O/prof_f16
