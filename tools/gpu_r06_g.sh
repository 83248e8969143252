#!/bin/bash
# round 6, call g: ResNet-50 pullback on the main stream (1, default) vs on a third stream (2) vs the side stream (0)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_resnet.py -m gpu -x -q -k "schedule_switches" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log | cut -c1-220
for rep in 1 2 3; do
for v in 1 2 0; do
XMC_RESNET_BWD_MAIN=$v timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_bwd_main_$v.txt
done
done
