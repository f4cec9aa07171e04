#!/bin/bash
# the conv data gradients on the streaming loop (operand instance) on / off: student step, same box, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5op; O=gpurun_out/r5op/log.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "gemm or engine or grad" >> $O 2>&1
for r in 1 2 3; do for v in 0 1; do
  echo "opnd=$v: $(FH_GEMM_STREAM_OPND=$v timeout 300 python tools/student_step.py --batch 256 --steps 10 --warmup 3 2>&1 | grep 'student step')" >> $O
done; done
tail -12 $O
