#!/bin/bash
# same-box A/B of library builds: tools/scratch/ab.sh "<command printing one line>" [rounds] [variants...]   (tools/scratch/ab_libs/lib_<variant>.so, untracked)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
CMD=$1; R=${2:-2}; shift; shift
V=${@:-base new}
L=fithubert_amd/libfithubert_hip.so
cp $L /tmp/lib_keep.so
for r in $(seq $R); do for v in $V; do
  cp tools/scratch/ab_libs/lib_$v.so $L
  echo "$v: $(bash -c "$CMD" 2>/dev/null | tail -1)"
done; done
cp /tmp/lib_keep.so $L
