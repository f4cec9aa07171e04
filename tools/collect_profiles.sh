#!/bin/bash
# Copy the summaries tools/final_measure.sh left in gpurun_out/ into profiles/ (tracked), round-tagged.
R=${1:-r05}
cd "$(dirname "$0")/.."
# attribution header for every summary: the commit checked out when the measurement set was collected (the GPU box has no .git) and the
# hashes of the sources / library that were measured (fithubert_amd/build.py; the same hashes are inside pmc_traffic.json and the bench line)
HEAD=$(git rev-parse HEAD 2>/dev/null || echo unknown)
SRC=$(python -c "from fithubert_amd.build import source_hash; print(source_hash())")
LIBH=$(python -c "from fithubert_amd.build import lib_hash; print(lib_hash())")
ATTR="# measured tree: git HEAD $HEAD (+ uncommitted edits if any), csrc source_sha256 $SRC, libfithubert_hip.so sha256 $LIBH"
tail -1 gpurun_out/bench_default.json > profiles/${R}_bench_default.json
tail -1 gpurun_out/benchB_under_rocprof.json > profiles/${R}_bench_B256_under_rocprof.json
{ echo "$ATTR"; echo "# FH_GEMM_TABLE=gpurun_out/gemm_table.json rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-isolated"
  echo "# (table written by a preceding un-profiled run, so no tuning launches; B=256 x 10 s per GPU, bf16, teacher on a side stream; 10 steps + 1 prefetch teacher forward; tools/final_measure.sh)"
  cat gpurun_out/benchB_kernel_stats.csv; } > profiles/${R}_bench_B256_kernel_stats.csv
python - <<PY
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
d["git_head_at_collection"] = "$HEAD"
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
python tools/merge_traffic.py profiles/${R}_bench_default.json
[ -f gpurun_out/student_kernel_stats.csv ] && { echo "$ATTR"; echo "# rocprofv3 --kernel-trace --stats -- python tools/student_step.py --batch 256 --steps 8 --warmup 2 (student only: forward + KD loss + backward; GEMM table reused; tools/prof_student.sh)"; cat gpurun_out/student_kernel_stats.csv; } > profiles/${R}_student_step_B256_kernel_stats.csv
[ -f gpurun_out/teacher_kernel_stats.csv ] && { echo "$ATTR"; echo "# rocprofv3 --kernel-trace --stats -- python tools/teacher_step.py --native 1 --batch 256 --steps 8 --warmup 2 (teacher forward only, every kernel native; GEMM table reused from an un-profiled run: no tuning launches in the trace; tools/prof_teacher.sh)"; cat gpurun_out/teacher_kernel_stats.csv; } > profiles/${R}_teacher_fwd_B256_kernel_stats.csv
ls -la profiles/
