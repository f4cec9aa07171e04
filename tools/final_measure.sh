#!/bin/bash
# The measurement set behind profiles/ and DESIGN.md section 5, in one call on the GPU box:
#   bash tools/final_measure.sh   (outputs under gpurun_out/; copy the summaries to profiles/ with tools/collect_profiles.sh)
# 1. an un-profiled run writes the per-shape GEMM table, so that profiled runs launch no tuning candidates
# 2. the default bench line (with the CPU baseline leg)
# 3. rocprofv3 --kernel-trace --stats of the same workload (table reused, no isolated pass: the kernel averages then match
#    the live HIP-event figures of that run)
# 4. three rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) and a per-launch
#    GEMM trace -> per-launch HBM traffic, MFMA-busy and algorithmic bytes / FLOPs of the GEMM kernels (tools/pmc_summary.py)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
T=gpurun_out/gemm_table.json; rm -f $T
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated --no-variants --gemm-table $T > /dev/null 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
FH_GEMM_TABLE=$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python bench.py --steps 8 --warmup 2 --prof-steps 8 --no-kd-delta --no-cpu-baseline --no-isolated --no-variants > gpurun_out/benchB_under_rocprof.json 2> gpurun_out/rocprof.err
cp $(ls gpurun_out/prof_final/*/*kernel_stats.csv | head -1) gpurun_out/benchB_kernel_stats.csv
FH_GEMM_TABLE=$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --no-variants --prof-steps 0 > /dev/null 2>&1
FH_GEMM_TABLE=$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --no-variants --prof-steps 0 > /dev/null 2>&1
FH_GEMM_TABLE=$T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated --no-variants --prof-steps 0 > /dev/null 2>&1
rm -f gpurun_out/gemm_trace_final.txt
FH_GEMM_TABLE=$T FH_GEMM_TRACE=gpurun_out/gemm_trace_final.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated --no-variants --prof-steps 1 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write ${FH_BENCH_BATCH:-256} gpurun_out/pmc_sq gpurun_out/gemm_trace_final.txt > gpurun_out/pmc_traffic.json 2> gpurun_out/pmc.err
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
tail -1 gpurun_out/bench_default.json | cut -c1-400
tail -1 gpurun_out/benchB_under_rocprof.json | cut -c1-300
