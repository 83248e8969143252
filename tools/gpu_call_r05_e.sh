#!/bin/bash
# round 5, call E: the optimiser kernel emits the prepared weights (xmc_adam_wprep_tiles) -- tests, then the same-box A/B
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_fused_opt.py tests/test_gpu_graph.py tests/test_gpu_step.py tests/test_gpu_dp.py > $O/tests.log 2>&1
tail -12 $O/tests.log
bash tools/ab_env.sh XMC_FUSE_PREP 2>&1 | tee $O/ab_fuse_prep.txt
