#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FH_GEMM_TABLE=gpurun_out/ex_table.json; rm -f $FH_GEMM_TABLE
python tools/scratch/ex_step.py 2>/dev/null | tail -1
rm -rf gpurun_out/prof_ex
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ex -- python tools/scratch/ex_step.py > gpurun_out/ex_prof.log 2>&1
cp $(ls gpurun_out/prof_ex/*/*kernel_stats.csv | head -1) gpurun_out/ex_kernel_stats.csv; rm -rf gpurun_out/prof_ex
tail -1 gpurun_out/ex_prof.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/ex_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6/12
print('kernel ms per step', tot)
for r in rows[:26]: print('  ', r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1.2e7,3), round(float(r['AverageNs'])/1e3,1))
PY
