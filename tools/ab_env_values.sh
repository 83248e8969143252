# same-box A/B of an env variable over several VALUES: alternated bench runs (G/D-only line included)
# usage: bash tools/ab_env_values.sh XMC_KSPLIT_TARGET 640 256 384
SW=$1; shift
for r in 1 2; do
  for v in "$@"; do
    env $SW=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$SW=$v', 'step', d['ms_per_step'], 'gd_only', d.get('gd_only',{}).get('ms_per_step'), {k: round(v,3) for k,v in d['losses'].items()})"
  done
done
