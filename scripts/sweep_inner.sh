for cfg in "1 1" "1 0" "2 1" "3 1"; do
  set -- $cfg
  echo "== inner=$1 lookahead=$2"
  XMCA_JACOBI_INNER=$1 XMCA_JACOBI_LOOKAHEAD=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rule-n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'])"
done
