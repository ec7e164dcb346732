for cfg in "1" "0"; do
  echo "== cross=$cfg"
  XMCA_JACOBI_CROSS=$cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rule-n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages_ms'], d['self_check'])"
done
