#!/bin/bash
# round 5, call K: phase weight gradients -- single-block last tile (tests, A/B), workgroup target sweep around the new default
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "wgrad" > $O/tests_k.log 2>&1
tail -3 $O/tests_k.log
timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_step.py -k "reproducible or c1_shapes or benchmarked_workload_c1" > $O/tests.log 2>&1
tail -3 $O/tests.log
bash tools/ab_env.sh XMC_WGRAD_NB1 2>&1 | tee $O/ab_wgrad_nb1.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_PHASE 512 384 640 768 2>&1 | tee $O/sweep_wgrad_phase2.txt
PYTHONPATH=$R timeout 900 python tools/bench_phase.py 2>&1 | grep -v amdgpu > $O/r05_conv_phase_vs_3x3.txt
tail -14 $O/r05_conv_phase_vs_3x3.txt
