#!/bin/bash
# round 5, call B: first-write gradients -- the optimiser / step / graph / replica tests, then the same-box A/B
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_fused_opt.py tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_gpu_dp.py \
  tests/test_gpu_kernels.py -k "not mx8" > $O/tests.log 2>&1
tail -8 $O/tests.log
bash tools/ab_env.sh XMC_FIRST_WRITE 2>&1 | tee $O/ab_first_write.txt
bash tools/ab_env.sh XMC_XC_REAL_HALF 2>&1 | tee $O/ab_xc_real_half.txt
bash tools/ab_env.sh XMC_GB_CONTIG 2>&1 | tee $O/ab_gb_contig.txt
