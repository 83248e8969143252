# MX scale without a saturating block maximum: the fp8 suite (incl. the bias-over-seeds test, printed) and C3 / C4
R=$PWD; O=$R/gpurun_out/am; mkdir -p $O
export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_gpu_mx8.py -x -q -m gpu -s 2>&1 | grep -v "amdgpu\|Warning\|warn" | tail -25 | tee $O/tests_mx8_scale.txt
run() { timeout 400 python bench.py --config $1 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$2', 'ms/step', d['ms_per_step'], 'img/s', d['value'], {k: round(v,3) for k,v in d['losses'].items()})"; }
run c3 "C3 bf16" | tee $O/c3c4.txt; run c4 "C4 fp8 " | tee -a $O/c3c4.txt
