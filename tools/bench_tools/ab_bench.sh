#!/bin/bash
# ab_bench.sh "<ENV=VALUE ...>": bench.py (8 steps) on one box with and without the given environment override, two rounds
cd "$(dirname "$0")/../.."
run() { timeout -k 10 300 env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  echo -n "default:  "; run FMI_AB_DUMMY=1
  echo -n "override: "; run $1
done
