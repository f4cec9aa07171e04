#!/bin/bash
# A/B: CUs left free by the persistent GEMM loop (loop 8) on one GPU; and loop 7 / 8 tie-break
for v in 0 8 16 32 0 8; do
  FH_RESERVE_CUS=$v python bench.py --steps 12 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reserve=$v', d['value'], d['ms_per_step'])"
done
for v in 0 4; do
  FH_GEMM_PREFER8=$v python bench.py --steps 12 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefer8=$v', d['value'], d['ms_per_step'])"
done
