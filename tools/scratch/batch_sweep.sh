#!/bin/bash
for b in 128 160 192 256 128; do
  python bench.py --batch $b --steps 10 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', d['value'], d['ms_per_step'])"
done
