#!/bin/bash
# same-box A/B of two builds of the library (tools/scratch/ab_libs/lib_old.so, lib_new.so; not tracked): the teacher's Linear shapes, then the bench
L=fithubert_amd/libfithubert_hip.so
for r in 1 2; do for v in old new; do
  cp tools/scratch/ab_libs/lib_$v.so $L
  echo "== $v"; python tools/scratch/lt_vs_own.py 2>/dev/null | tail -7 | awk '{print $1, $2, $(NF-9), $(NF-8), "loop8", $(NF-6), $(NF-3), $(NF-2), $(NF-1), $NF}'
done; done
for r in 1 2; do for v in old new; do
  cp tools/scratch/ab_libs/lib_$v.so $L
  python bench.py --steps 12 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v bench', d['value'], d['ms_per_step'])"
done; done
cp tools/scratch/ab_libs/lib_new.so $L
