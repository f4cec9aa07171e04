#!/bin/bash
# round 5: the full step with the weight gradients on the side stream (default) against serialised (FH_WGRAD_STREAM=0), alternating, same box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p3; mkdir -p $O
F="--steps 12 --warmup 3 --no-cpu-baseline --no-isolated --no-variants --no-kd-delta --prof-steps 0"
for rep in 1 2; do
  for w in 1 0; do
    FH_WGRAD_STREAM=$w python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_stream=$w', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
