#!/bin/bash
# round 6, call b: the new ResNet-50 launches (dual-source pointwise, fused stem, compact 3x3) -- tests, per-launch table, step A/B
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_mx8.py tests/test_gpu_dp.py -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -25 $O/pytest.log
XMC_RESNET_DUAL=0 XMC_RESNET_STEM_FUSED=0 XMC_RESNET_SKIP3=0 timeout 300 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/resnet_per_launch_r05_path.txt; tail -1 $O/resnet_per_launch_r05_path.txt
timeout 300 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/resnet_per_launch.txt; grep "^pass\|TOTAL\|stem_conv\|maxpool3x3s2 " $O/resnet_per_launch.txt
for rep in 1 2; do
XMC_RESNET_DUAL=0 XMC_RESNET_STEM_FUSED=0 XMC_RESNET_SKIP3=0 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_off.txt
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_on.txt
done
