#!/bin/bash
# same-box A/B: teacher forward alone (B=128) with its Linears on the own GEMM (default) vs hipBLASLt through F.linear
for i in 1 2; do
for v in native torch; do
  echo -n "$v: "; FH_TEACHER_LINEAR=$v python tools/teacher_step.py --native 1 --batch 128 --steps 10 --warmup 3 2>/dev/null | grep "teacher fwd"
done; done
python tools/scratch/lt_vs_own.py 2>/dev/null | tail -12
