#!/bin/bash
# teacher posconv (48-channel groups): Toeplitz tap order on / off (same library, env switch)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5pc; O=gpurun_out/r5pc/log8.txt; : > $O
timeout 900 python -m pytest tests -m gpu -x -q -k "posconv or teacher" >> $O 2>&1
for r in 1 2; do for v in 0 1; do
    echo "== wide toeplitz=$v D=768 B=256" >> $O
    FH_POSCONV_WIDE_TOEPLITZ=$v B=256 D=768 timeout 300 python tools/posconv_bench.py 2>&1 | grep wide >> $O
done; done
cat $O
