#!/bin/bash
# kernel stats of the fp16 bench beside the bf16 one (same box): where the 4-5 % go
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for dt in bf16 float16; do
  T=gpurun_out/table_$dt.json; rm -f $T
  python bench.py --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline --no-isolated --no-variants --gemm-table $T > /dev/null 2>&1
  rm -rf gpurun_out/prof_$dt
  FH_GEMM_TABLE=$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$dt -- python bench.py --dtype $dt --steps 8 --warmup 2 --no-cpu-baseline --no-isolated --no-variants --prof-steps 0 2>/dev/null | cut -c1-150
  cp $(ls gpurun_out/prof_$dt/*/*kernel_stats.csv | head -1) gpurun_out/kstats_$dt.csv
  rm -rf gpurun_out/prof_$dt
done
