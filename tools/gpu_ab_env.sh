#!/bin/bash
# same-box A/B of one environment switch: bash tools/gpu_ab_env.sh VAR [rounds] -- per-layer conv table + step with VAR=0 / VAR=1
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
V=$1; N=${2:-2}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "conv_stream_packed or conv_phase or mask_bits" 2>&1 | tail -3
for r in $(seq $N); do for x in 0 1; do
  echo "== $V=$x (round $r)"
  env $V=$x timeout 300 python tools/bench_conv.py --packed ${BENCH_CONV_ARGS:-} 2>&1 | grep -v amdgpu | grep -E "${ROWS:-96|192|TOTAL|layer}"
  env $V=$x timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gd_only', d['gd_only']['ms_per_step'])"
done; done
