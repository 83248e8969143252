#!/bin/bash
# same-box A/B of one environment switch on the phase kernels: bash tools/gpu_ab_phase_env.sh VAR [rounds]
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
V=$1; N=${2:-2}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_phase or stride2 or mask_bits" 2>&1 | tail -3
for r in $(seq $N); do for x in 0 1; do
  echo "== $V=$x (round $r)"
  env $V=$x timeout 300 python tools/bench_phase.py --only-phase --iters 3 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,"| fwd",$9,$10,"| dgrad",$18,$19}'
  env $V=$x timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gd_only', d['gd_only']['ms_per_step'])"
done; done
