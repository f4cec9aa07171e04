#!/bin/bash
# A/B: teacher on a side stream (default) vs the student's stream, at two batch sizes
for s in 1 0; do for b in 128 16; do
  FH_TEACHER_STREAM=$s timeout 280 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-isolated 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$s" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("side_stream", sys.argv[1], "B", d["config"].get("per_gpu_batch"), "utt/s", d["value"], "ms", d["ms_per_step"], "gemm TF/s", d["roofline"]["achieved"], "frac", d["roofline"]["frac"])
PY
done; done
