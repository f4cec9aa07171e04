for v in native torch; do
  FH_TEACHER_LINEAR=$v python bench.py --batch 3 --steps 30 --warmup 5 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=3', d['value'], d['ms_per_step'])"
  FH_TEACHER_LINEAR=$v python bench.py --batch 16 --steps 30 --warmup 5 --no-variants --no-cpu-baseline --no-isolated 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v B=16', d['value'], d['ms_per_step'])"
done
