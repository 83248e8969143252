# same-box A/B of an env switch: alternated bench runs (G/D-only line included)
SW=$1; shift
for r in 1 2; do
  for v in 0 1; do
    env $SW=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$SW=$v', 'step', d['ms_per_step'], 'gd_only', d.get('gd_only',{}).get('ms_per_step'), {k: round(v,3) for k,v in d['losses'].items()})"
  done
done
