# config #5 (MX-fp8): which layers should take the fp8 kernel?  C4 at several channel thresholds next to C3, alternated, one box
R=$PWD; O=$R/gpurun_out/w; mkdir -p $O
export PYTHONPATH=$R
run() { timeout 400 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', 'ms/step', d['ms_per_step'], 'img/s', d['value'], {k: round(v,3) for k,v in d['losses'].items()})"; }
for r in 1 2; do
  run c3 "C3 bf16            "
  for m in 64 256 512 1024; do XMC_FP8_MIN_CIN=$m run c4 "C4 fp8 cin >= $m"; done
done 2>&1 | tee $O/c4_min_cin.txt
