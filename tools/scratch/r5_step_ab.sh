#!/bin/bash
# same-box A/B of the student step and the teacher forward: round-start library (r5), + posconv work (pc), + fused q|k|v bias gradient (qb)
# (the older libraries lack fh_attention_backward_bias: on this scratch copy of the tree the ctypes table drops the entry for them)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5ab; O=gpurun_out/r5ab/log.txt; : > $O
L=fithubert_amd/libfithubert_hip.so; cp $L /tmp/lib_keep.so; cp fithubert_amd/_lib.py /tmp/_lib_keep.py
run() { # variant, command
  cp tools/scratch/ab_libs/lib_$1.so $L; cp /tmp/_lib_keep.py fithubert_amd/_lib.py
  [ "$1" != qb ] && sed -i '/fh_attention_backward_bias/d' fithubert_amd/_lib.py
  echo "$1: $(bash -c "$2" 2>&1 | tail -1)" >> $O
}
echo "--- student step B=256" >> $O
for r in 1 2 3; do for v in r5 pc qb; do run $v "timeout 300 python tools/student_step.py --batch 256 --steps 10 --warmup 3 | grep 'student step'"; done; done
echo "--- teacher forward B=256" >> $O
for r in 1 2 3; do for v in r5 qb; do run $v "timeout 300 python tools/teacher_step.py --native 1 --batch 256 --steps 10 --warmup 3 | grep 'teacher fwd'"; done; done
cp /tmp/lib_keep.so $L; cp /tmp/_lib_keep.py fithubert_amd/_lib.py
cat $O
