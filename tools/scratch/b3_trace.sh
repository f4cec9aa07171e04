#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=gpurun_out/b3_table.json; rm -f $T
python bench.py --batch 3 --steps 10 --warmup 5 --no-cpu-baseline --no-isolated --no-variants --gemm-table $T 2>/dev/null | cut -c1-200
rm -rf gpurun_out/b3_tl
FH_GEMM_TABLE=$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b3_tl -- python bench.py --batch 3 --steps 12 --warmup 4 --no-cpu-baseline --no-isolated --no-variants --prof-steps 0 > gpurun_out/b3_under.json 2>/dev/null
cut -c1-200 gpurun_out/b3_under.json
python tools/timeline_window.py $(ls gpurun_out/b3_tl/*/*kernel_trace.csv | head -1) 3
python tools/timeline_window.py $(ls gpurun_out/b3_tl/*/*kernel_trace.csv | head -1) 5
cp $(ls gpurun_out/b3_tl/*/*kernel_stats.csv | head -1) gpurun_out/b3_kernel_stats.csv
rm -rf gpurun_out/b3_tl
