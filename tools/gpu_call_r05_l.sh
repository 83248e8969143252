#!/bin/bash
# round 5, call L: contrastive_loss in two launches per direction (tests + A/B); lower split targets for the weight gradients
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "contrastive or xent or wgrad_phase" > $O/tests_k.log 2>&1
tail -3 $O/tests_k.log
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py tests/test_gpu_resnet.py tests/test_gpu_graph.py > $O/tests.log 2>&1
tail -4 $O/tests.log
bash tools/ab_env.sh XMC_CL_FUSED 2>&1 | tee $O/ab_cl_fused.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_PHASE 384 256 320 2>&1 | tee $O/sweep_wgrad_phase3.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_HI 384 256 320 2>&1 | tee $O/sweep_wgrad_hi2.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_LO 512 256 384 2>&1 | tee $O/sweep_wgrad_lo2.txt
