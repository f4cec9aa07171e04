#!/bin/bash
# micro-benchmarks behind DESIGN.md's round-3 claims, one log each (copied to profiles/)
python tools/attn_bench.py 2>/dev/null > gpurun_out/ev_attn_bench.txt
python tools/scratch/relation_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/ev_relation_bench.txt
python tools/scratch/libri_diag.py 2>/dev/null | grep -v amdgpu > gpurun_out/ev_libri_diag.txt
python tools/scratch/lt_vs_own.py 2>/dev/null | grep -v amdgpu > gpurun_out/ev_lt_vs_own.txt
bash tools/scratch/teacher_ab.sh 2>/dev/null | head -4 > gpurun_out/ev_teacher_ab.txt
python tools/scratch/small_gemm_lat.py 2>/dev/null | grep -v amdgpu > gpurun_out/ev_small_gemm.txt
ls gpurun_out/ev_*.txt
