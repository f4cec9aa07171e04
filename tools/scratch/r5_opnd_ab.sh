#!/bin/bash
# the conv data gradients on the streaming loop (operand instance) on / off: whole bench step, same box, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5op; O=gpurun_out/r5op/bench_ab.txt; : > $O
for r in 1 2 3; do for v in 0 1; do
  echo "opnd=$v: $(FH_GEMM_STREAM_OPND=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-isolated --no-variants --no-kd-delta --prof-steps 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" >> $O
done; done
cat $O
