#!/bin/bash
# round-6 microbenchmarks on the final binary: attention, own loops vs hipBLASLt on the teacher's Linear shapes, LayerNorm, conv 0, positional conv
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6micro; mkdir -p $O
python tools/attn_bench.py 2>/dev/null > $O/attn.txt
B=256 python tools/attn_bench.py 2>/dev/null | head -3 > $O/attn256.txt
python tools/scratch/lt_vs_own.py 2>/dev/null | grep -v amdgpu > $O/lt_vs_own.txt
python tools/ln_bench.py 2>/dev/null | grep -v amdgpu > $O/ln.txt
for c in 128 512; do for m in 1 0; do echo "FH_CONV0_MFMA=$m $(FH_CONV0_MFMA=$m B=256 C0=$c python tools/scratch/conv0_bench.py 2>/dev/null | head -1)"; done; done > $O/conv0.txt
python tools/scratch/pc_ab.py 2>/dev/null > $O/posconv.txt
cat $O/attn.txt $O/attn256.txt $O/lt_vs_own.txt $O/ln.txt $O/conv0.txt $O/posconv.txt
