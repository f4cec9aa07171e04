cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/scratch/stream_bench.py 2>&1 | grep -v amdgpu.ids
