cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_teacher_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
python tools/student_step.py --batch 128 --steps 20 --warmup 3 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-variants --no-isolated 2>/dev/null | tail -1 | cut -c1-200
