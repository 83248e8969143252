#!/bin/bash
# round 5, call J: split-K workgroup targets re-swept inside the step after first-write gradients (the reducing passes got cheaper)
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
export PYTHONPATH=$R
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_HI 384 512 768 2>&1 | tee $O/sweep_wgrad_hi.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_LO 512 768 384 2>&1 | tee $O/sweep_wgrad_lo.txt
bash tools/ab_env_values.sh XMC_WGRAD_TARGET_PHASE 768 1024 512 2>&1 | tee $O/sweep_wgrad_phase.txt
bash tools/ab_env_values.sh XMC_KSPLIT_TARGET 256 384 2>&1 | tee $O/sweep_ksplit.txt
bash tools/trace_gd.sh > $O/trace_gd.out 2>&1
cp gpurun_out/tr_gd/stats_gd.txt $O/r05_rocprofv3_kernel_trace_stats_bench_gd_only.txt
cp gpurun_out/tr_gd/timeline_gd.txt $O/r05_timeline_gd_only.txt
head -30 $O/r05_timeline_gd_only.txt
