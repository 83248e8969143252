#!/bin/bash
# round 6, call d: chained pointwise launches (xmc_conv2d_pw_chain) -- tests, per-launch table, step A/B; XMC_RESNET_REAL_EARLY A/B
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_resnet.py -m gpu -x -q -k "chain or dual or stem or folded or graphed" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -16 $O/pytest.log | cut -c1-220
XMC_RESNET_CHAIN=0 timeout 300 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/resnet_per_launch_chain_off.txt; grep "^pass 3\|TOTAL" $O/resnet_per_launch_chain_off.txt
timeout 300 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/resnet_per_launch.txt; grep "^pass 3\|TOTAL" $O/resnet_per_launch.txt
for rep in 1 2; do
XMC_RESNET_CHAIN=0 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_chain_off.txt
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_on.txt
XMC_RESNET_REAL_EARLY=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_on_real_early.txt
done
