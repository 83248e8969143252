#!/bin/bash
# round 5, call I: two-stage ring on the 4x4-map weight gradients; attention context written into / read out of the spatial condition
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "wgrad or attention" > $O/tests_k.log 2>&1
tail -3 $O/tests_k.log
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py tests/test_gpu_graph.py > $O/tests.log 2>&1
tail -5 $O/tests.log
timeout 600 python tools/bench_conv.py --packed --wgrad-tunes 1,8193 --wgrad-raw --only " 4 " 2>&1 | grep -v amdgpu | tee $O/wgrad_4x4_nst.txt
timeout 600 python tools/bench_conv.py --packed --wgrad-tunes 1,8193 --wgrad-raw --only "4>8" 2>&1 | grep -v amdgpu | tee -a $O/wgrad_4x4_nst.txt
bash tools/ab_env.sh XMC_WGRAD_NST3 2>&1 | tee $O/ab_wgrad_nst3.txt
bash tools/ab_env.sh XMC_SCOND_DIRECT 2>&1 | tee $O/ab_scond_direct.txt
timeout 600 python tools/torch_kernel_sites.py --pretrained off 2>&1 | grep -v amdgpu > $O/r05_torch_op_sites_gd_only.txt
timeout 600 python tools/torch_kernel_sites.py 2>&1 | grep -v amdgpu > $O/r05_torch_op_sites.txt
head -40 $O/r05_torch_op_sites_gd_only.txt
