#!/bin/bash
# round 5, call P: non-temporal epilogue stores (variant library, alternated with the shipped one); forward / data-gradient split targets
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
export PYTHONPATH=$R
bash tools/ab_lib.sh xmcgan_image_generation_amd/csrc/build_nt/libxmcgan_hip.so 3 python tools/bench_step_short.py --no-instrument 2>&1 | tee $O/ab_epilogue_nt_stores.txt
bash tools/ab_env_values.sh XMC_KSPLIT_TARGET_PHASE 384 256 512 2>&1 | tee $O/sweep_ksplit_phase.txt
bash tools/ab_env_values.sh XMC_KSPLIT_TARGET_PW 256 128 384 2>&1 | tee $O/sweep_ksplit_pw.txt
