#!/bin/bash
# same-box A/B of library builds: tools/scratch/ab.sh "<command printing one line>" [rounds] [variants...]
# (tools/scratch/ab_libs/lib_<variant>.so, untracked).  The variant is selected through FH_LIB_PATH (fithubert_amd/_lib.py reads it):
# the in-tree library is never overwritten, so an interrupted run leaves the tree as it was.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
CMD=$1; R=${2:-2}; shift; shift
V=${@:-base new}
for r in $(seq $R); do for v in $V; do
  echo "$v: $(FH_LIB_PATH=$PWD/tools/scratch/ab_libs/lib_$v.so bash -c "$CMD" 2>/dev/null | tail -1)"
done; done
