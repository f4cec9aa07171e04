#!/bin/bash
# A/B of the fused conv-0 backward (weight-gradient sums inside layer 1's data-gradient GEMM)
python -m pytest tests/test_ops_gpu.py -m gpu -q -k conv0 2>&1 | tail -4
python -m pytest tests/test_engine_gpu.py -m gpu -q 2>&1 | tail -4
for v in 0 1 0 1; do
  FH_CONV0_FUSED_BWD=$v python bench.py --steps 12 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$v', d['value'], d['ms_per_step'])"
done
