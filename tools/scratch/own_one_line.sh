#!/bin/bash
python tools/scratch/lt_vs_own.py 2>/dev/null | grep -v amdgpu | awk '{printf "%s %s L7 %s L8 %s | ", $1, $2, $(NF-11), $(NF-9)}'; echo
