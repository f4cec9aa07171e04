#!/bin/bash
# same-box A/B of two library builds (tools/scratch/ab_libs/lib_old.so, lib_new.so): student step alone and the bench, alternating
L=fithubert_amd/libfithubert_hip.so
for r in 1 2 3; do for v in old new; do
  cp tools/scratch/ab_libs/lib_$v.so $L
  echo -n "$v "; python tools/student_step.py --batch 128 --steps 12 --warmup 3 2>/dev/null | grep "student step"
done; done
for r in 1 2 3; do for v in old new; do
  cp tools/scratch/ab_libs/lib_$v.so $L
  python bench.py --steps 12 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', d['value'], d['ms_per_step'])"
done; done
cp tools/scratch/ab_libs/lib_new.so $L
