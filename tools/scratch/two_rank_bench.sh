#!/bin/bash
# bench.py's world-size-2 path on ONE GPU: two processes, collectives through gloo, both ranks on cuda:0 (LOCAL_RANK=0)
cd "$GRAFT_REPO_ROOT"
export FH_DIST_BACKEND=gloo WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-isolated > gpurun_out/r2_rank1.log 2> gpurun_out/r2_rank1.err &
RANK=0 python bench.py --gpus 2 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-isolated 2> gpurun_out/r2_rank0.err | tail -1 | cut -c1-330
wait
