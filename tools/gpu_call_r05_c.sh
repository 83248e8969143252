#!/bin/bash
# round 5, call C: 96-cout tiles of the weight-gradient kernel -- kernel tests, per-layer A/B, in-step A/B
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "wgrad" > $O/tests.log 2>&1
tail -5 $O/tests.log
timeout 600 python tools/bench_conv.py --packed --wgrad-tunes 1,2049 --wgrad-raw 2>&1 | tee $O/wgrad_c96_per_layer.txt
bash tools/ab_env.sh XMC_WGRAD_C96 2>&1 | tee $O/ab_wgrad_c96.txt
