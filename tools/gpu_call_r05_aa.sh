# config #5 with the fp8 kernel's new epilogue features (ReLU on store, bit masks): tests, then C3 vs C4 alternated
R=$PWD; O=$R/gpurun_out/w; mkdir -p $O
export PYTHONPATH=$R
timeout 1200 python -m pytest tests/test_gpu_mx8.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests_mx8_bits.txt
run() { timeout 400 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', 'ms/step', d['ms_per_step'], 'img/s', d['value'], {k: round(v,3) for k,v in d['losses'].items()})"; }
for r in 1 2; do
  run c3 "C3 bf16                              "
  run c4 "C4 fp8, ReLU stored in every block   "
  XMC_FP8_RELU_STORED=0 run c4 "C4 fp8, ReLU stored in bf16 blocks   "
  XMC_FP8_PHASE=0 run c4 "C4 fp8 incl. resampling-adjacent     "
done 2>&1 | tee $O/c4_relu_stored_everywhere.txt
