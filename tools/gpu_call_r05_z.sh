# config #5 after giving the bf16-kernel blocks their stored ReLU back and skipping the unused 3x3 copies of the phase sites
R=$PWD; O=$R/gpurun_out/w; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_mx8.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_mx8.txt
run() { timeout 400 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', 'ms/step', d['ms_per_step'], 'img/s', d['value'], {k: round(v,3) for k,v in d['losses'].items()})"; }
for r in 1 2; do
  run c3 "C3 bf16            "
  for m in 64 256; do XMC_FP8_MIN_CIN=$m run c4 "C4 fp8 cin >= $m"; done
done 2>&1 | tee $O/c4_after_relu_stored.txt
