#!/bin/bash
# tools/disasm.sh <unit> [outdir]: gfx950 disassembly of one translation unit's device code (fithubert_amd/csrc/_obj/<unit>.hip.o)
set -e
B=/opt/rocm/lib/llvm/bin
U=$1; D=${2:-/tmp/fh_dis}; mkdir -p $D
$B/llvm-objcopy --dump-section=.hip_fatbin=$D/$U.fat fithubert_amd/csrc/_obj/$U.hip.o $D/$U.x.o
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$D/$U.fat --output=$D/$U.co --unbundle
$B/llvm-objdump -d $D/$U.co > $D/$U.s
echo $D/$U.s
