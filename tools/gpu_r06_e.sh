#!/bin/bash
# round 6, call e: the whole GPU suite with per-test durations; XMC_RESNET_REAL_EARLY A/B
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_e; mkdir -p $O; cd $R
nproc > $O/host.txt; uptime >> $O/host.txt
for rep in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_on.txt
XMC_RESNET_REAL_EARLY=1 timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160 | tee -a $O/ab_on_real_early.txt
done
timeout 1700 python -m pytest tests -m gpu -x -q --durations=60 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
uptime >> $O/host.txt
tail -70 $O/pytest.log | cut -c1-160
