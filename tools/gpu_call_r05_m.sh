#!/bin/bash
# round 5, call M: contrastive_loss kernels with more loads in flight -- tests + A/B
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "contrastive or xent" > $O/tests_k.log 2>&1
tail -3 $O/tests_k.log
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py -k "fp32_tiny or c1_shapes or reproducible" > $O/tests.log 2>&1
tail -3 $O/tests.log
bash tools/ab_env.sh XMC_CL_FUSED 2>&1 | tee $O/ab_cl_fused.txt
bash tools/ab_env.sh XMC_CL_FUSED 2>&1 | tee -a $O/ab_cl_fused.txt
