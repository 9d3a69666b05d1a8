cd $GRAFT_REPO_ROOT
for r in 1 2; do
  echo -n "default: "; timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo -n "cout32:  "; FMI_P3_MIN_COUT=32 FMI_P3_MIN_WORK=288 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
