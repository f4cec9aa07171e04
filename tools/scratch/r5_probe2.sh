#!/bin/bash
# round 5: per-launch GEMM trace of the student step with the weight gradients serialised on the main stream (FH_WGRAD_STREAM=0)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p2; mkdir -p $O
export FH_WGRAD_STREAM=0
T=$O/table.json; rm -f $T
FH_GEMM_TABLE=$T python tools/student_step.py --batch 256 --steps 6 > $O/student.tune.log 2>&1
rm -f $O/trace_student.txt
FH_GEMM_TABLE=$T FH_GEMM_TRACE=$O/trace_student.txt python tools/student_step.py --batch 256 --steps 8 > $O/student.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/student_step.py --batch 256 --steps 6 > $O/prof.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/student_serial_kernel_stats.csv; rm -rf $O/prof
tail -n 3 $O/student.log
