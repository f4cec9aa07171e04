#!/bin/bash
# bench.py's default (fixed 10 s) workload with two ranks sharing the one GPU over gloo: the N > 1 code path of the headline bench
cd "$GRAFT_REPO_ROOT"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 WORLD_SIZE=2 FH_DIST_BACKEND=gloo LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/two_rank_r1.json 2> gpurun_out/two_rank_r1.err &
RANK=0 python bench.py --gpus 2 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/two_rank_r0.json 2> gpurun_out/two_rank_r0.err
wait
tail -1 gpurun_out/two_rank_r0.json | cut -c1-600; tail -3 gpurun_out/two_rank_r1.err
