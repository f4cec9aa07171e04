#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_GEMM_TABLE=gpurun_out/large_table.json; rm -f $FH_GEMM_TABLE
python tools/teacher_step.py --native 1 --arch large --batch 64 2>/dev/null | tail -1
rm -rf gpurun_out/prof_large
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_large -- python tools/teacher_step.py --native 1 --arch large --batch 64 --steps 8 --warmup 2 > gpurun_out/large_prof.log 2>&1
cp $(ls gpurun_out/prof_large/*/*kernel_stats.csv | head -1) gpurun_out/large_kernel_stats.csv; rm -rf gpurun_out/prof_large
tail -1 gpurun_out/large_prof.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/large_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/10
print('kernel ms per forward', tot)
for r in rows[:14]: print('  ', r['Name'][:84], r['Calls'], round(float(r['TotalDurationNs'])/1e7,3), round(float(r['AverageNs'])/1e3,1))
PY
