#!/bin/bash
# round 6, call a: the whole GPU suite in the product optimiser mode + the default bench line + the ResNet per-launch table (baseline of the round)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench.err; tail -1 $O/bench_c1.json | cut -c1-400
timeout 600 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/resnet_per_launch.txt; tail -3 $O/resnet_per_launch.txt
