#!/bin/bash
# round 5, call G: config #5 -- MX-fp8 on the non-resampled 3x3 layers only, bf16 phase kernels on the resampling-adjacent ones
# (XMC_FP8_PHASE=1) vs the fp8 3x3 kernel everywhere (=0) vs the bf16 configuration C3, alternated; then the fp8 tests with =1
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
export PYTHONPATH=$R
run() { # label, env..., args
  local label=$1; shift
  env "$@" timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-instrument --config $CFG 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', 'ms/step', d['ms_per_step'], 'img/s', d['value'], {k: round(v,3) for k,v in d['losses'].items()})"
}
for r in 1 2; do
  CFG=c3 run "C3 bf16          " XMC_DUMMY=0
  CFG=c4 run "C4 fp8 3x3 only  " XMC_FP8_PHASE=0
  CFG=c4 run "C4 fp8 + bf16 phase" XMC_FP8_PHASE=1
done 2>&1 | tee $O/c4_vs_c3.txt
for r in 1 2; do
  CFG=c1 run "C1 bf16          " XMC_DUMMY=0
  CFG=c1 run "C1+fp8 3x3 only  " XMC_FP8_PHASE=0 XMC_BENCH_FP8=1
done 2>&1 | tee -a $O/c4_vs_c3.txt
XMC_FP8_PHASE=1 timeout 1800 python -m pytest -q -x -m gpu tests/test_gpu_mx8.py > $O/tests_mx8_phase1.log 2>&1
tail -5 $O/tests_mx8_phase1.log
