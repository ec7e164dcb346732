#!/bin/bash
# scripts/sweep_env.sh VAR "v1 v2 ..." [bench args]  - bench.py once per value of an environment switch (GPU box)
var=$1; vals=$2; shift 2
for v in $vals; do
  echo -n "$var=$v  "
  env $var=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rule-n --no-e2e "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('ms_per_step %.2f  eigh %.2f  round %.2f us  frac %.3f  sweeps %d' % (d['ms_per_step'], d['stages_ms']['eigh'], r['avg_launch_us'], r['frac'], d['stages_ms']['eigh_info']['sweeps']))"
done
