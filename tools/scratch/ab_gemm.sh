#!/bin/bash
# A/B of two library builds on the own-loop GEMM microbenchmarks and the teacher forward
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do for v in $@; do
  echo "== $v"; FH_LIB_PATH=tools/scratch/ab_libs/lib_$v.so python tools/scratch/lt_vs_own.py 2>/dev/null | awk '{print $1, $2, $3, "loop7", $(NF-11), "loop8", $(NF-9)}' | tr '\n' ';'; echo
done; done
