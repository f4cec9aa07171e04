#!/bin/bash
# round-5 microbenchmarks on the final binary: attention, own loops vs hipBLASLt on the teacher's Linear shapes, batch sweep of the bench
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5micro; mkdir -p $O
python tools/attn_bench.py 2>/dev/null > $O/attn.txt
python tools/scratch/lt_vs_own.py 2>/dev/null | grep -v amdgpu > $O/lt_vs_own.txt
python tools/ln_bench.py 2>/dev/null | grep -v amdgpu > $O/ln.txt
for b in 256 384 512 256; do
  python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-isolated --no-variants --no-kd-delta --prof-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', d['value'], 'utt/s', d['ms_per_step'], 'ms/step')" >> $O/batch.txt
done
cat $O/attn.txt $O/lt_vs_own.txt $O/ln.txt $O/batch.txt
