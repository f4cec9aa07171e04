#!/bin/bash
# Does restricting a stream's GEMM loops to the small-LDS ones (fh_gemm_stream_loops) buy real co-residence of the teacher's and the
# student's kernels?  Same box, same binary: bench.py --steps 10 with FH_TEACHER_LOOPS / FH_STUDENT_LOOPS combinations.
out=${1:-gpurun_out/overlap_probe.txt}
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --steps 10 --warmup 3 --no-variants --no-cpu-baseline --no-isolated 2>>$out.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $out
}
run FH_NOP=1
run FH_TEACHER_LOOPS=2,4
run FH_TEACHER_LOOPS=2,4,1
run FH_STUDENT_LOOPS=1,2,4
run FH_TEACHER_LOOPS=2,4 FH_STUDENT_LOOPS=1,2,4
run FH_TEACHER_STREAM=0
run FH_TEACHER_LINEAR=native
cat $out
