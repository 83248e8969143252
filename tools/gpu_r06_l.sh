#!/bin/bash
# round 6, call l: the ResNet-50 forward behind the prefetched generator forward (XMC_RESNET_FWD_PREFETCH) vs the default, vs REAL_EARLY
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_l; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_resnet.py -m gpu -x -q -k "schedule_switches" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log | cut -c1-220
for rep in 1 2 3; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_default.txt
XMC_RESNET_FWD_PREFETCH=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_fwd_prefetch.txt
XMC_RESNET_REAL_EARLY=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_real_early.txt
done
