# one sample of the box-to-box spread: the default bench line (both workloads) of the current tree on whatever box this call got
R=$PWD; O=$R/gpurun_out/boxes; mkdir -p $O
export PYTHONPATH=$R
T=$(date +%H%M%S)
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$T.json
python -c "
import json,sys
d=json.loads(open('$O/bench_$T.json').read())
print('box sample $T: step', d['ms_per_step'], 'img/s', d['value'], 'gd_only', d['gd_only']['ms_per_step'], 'frac', d['roofline']['frac'], 'wgrad', d['roofline']['wgrad']['frac'])"
