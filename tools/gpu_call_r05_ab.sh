#!/bin/bash
# round 5, call AB (final, after the fused global projections, the config-5 epilogue work and the input-pipeline work): the whole GPU suite + smoke, the round's profile set regenerated on the final code, G/D-only trace + timeline,
# torch op sites
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05ab
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r05 2>&1 | tail -12
bash tools/trace_gd.sh > $O/trace_gd.out 2>&1
cp gpurun_out/tr_gd/stats_gd.txt gpurun_out/prof_r05/r05_rocprofv3_kernel_trace_stats_bench_gd_only.txt
cp gpurun_out/tr_gd/timeline_gd.txt gpurun_out/prof_r05/r05_timeline_gd_only.txt
timeout 600 python tools/torch_kernel_sites.py --pretrained off 2>&1 | grep -v amdgpu > gpurun_out/prof_r05/r05_torch_op_sites_gd_only.txt
timeout 600 python tools/torch_kernel_sites.py 2>&1 | grep -v amdgpu > gpurun_out/prof_r05/r05_torch_op_sites.txt
PYTHONPATH=$R timeout 900 python tools/bench_conv.py --packed --wgrad-tunes 1,2049 --wgrad-raw 2>&1 | grep -v amdgpu > gpurun_out/prof_r05/r05_wgrad_c96_per_layer.txt
