#!/bin/bash
# round 5, call D: schedule experiments -- prefetched generator forward from the end of D's trunk; D's heads on two streams
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_fused_opt.py tests/test_cabi_c.py > $O/tests.log 2>&1
tail -5 $O/tests.log
bash tools/ab_env.sh XMC_PREFETCH_EARLY 2>&1 | tee $O/ab_prefetch_early.txt
bash tools/ab_env.sh XMC_HEADS_2STREAM 2>&1 | tee $O/ab_heads_2stream.txt
