#!/bin/bash
# positional conv kernels of the current build: parity + timing at the student's and the teacher's shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5pc; O=gpurun_out/r5pc/log9.txt; : > $O
timeout 900 python -m pytest tests -m gpu -x -q -k "posconv" >> $O 2>&1
for r in 1 2; do
  B=256 D=480 timeout 300 python tools/posconv_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  B=256 D=768 timeout 300 python tools/posconv_bench.py 2>&1 | grep wide >> $O
done
cat $O
