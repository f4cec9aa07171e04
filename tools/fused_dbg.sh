#!/bin/bash
# timing ablations of the fused layer kernels (FH_FUSED_DBG bits: 1 no weight fetch, 2 no MFMA, 4 no global stores/fills, 8 no LDS commit)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for d in 0 1 2 4 8 3 7 15; do
  echo "dbg=$d"
  FH_FUSED_DBG=$d FH_WGRAD_STREAM=0 FH_GEMM_TABLE=gpurun_out/gemm_table_s.json rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -- python tools/student_step.py --batch 128 --steps 3 --warmup 1 > /dev/null 2>&1
  grep -E "layer_rows" $(ls gpurun_out/prof_d/*/*kernel_stats.csv | head -1) | cut -d, -f1-4 | cut -c40-
  rm -rf gpurun_out/prof_d
done
