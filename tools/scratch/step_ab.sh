#!/bin/bash
# student step / teacher forward / bench line of library variants, same box, alternating: tools/scratch/step_ab.sh rounds variant[:ENV=VAL[,ENV2=VAL2]]...
cd "$GRAFT_REPO_ROOT"; R=$1; shift
for r in $(seq $R); do for spec in $@; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  L=$PWD/tools/scratch/ab_libs/lib_$v.so
  s=$(env $envs FH_LIB_PATH=$L python tools/student_step.py --batch 256 --steps 10 --warmup 3 2>/dev/null | grep -o "student step B=256: [0-9.]* ms")
  t=$(env $envs FH_LIB_PATH=$L python tools/teacher_step.py --native 1 --batch 256 --steps 10 --warmup 3 2>/dev/null | grep -o "teacher fwd B=256: [0-9.]* ms")
  b=$(env $envs FH_LIB_PATH=$L python bench.py --steps 12 --no-variants --no-cpu-baseline --no-kd-delta --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$spec: $s | $t | bench $b"
done; done
