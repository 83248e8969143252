#!/bin/bash
# round 6, call o: stem kernels with the XCD-aware tile order -- tests + per-launch times
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_o; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_resnet.py -m gpu -x -q -k "stem or benchmarked_size" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
for rep in 1 2; do timeout 300 python tools/bench_resnet.py --detail 2>&1 | grep "stem_\|TOTAL" | tee -a $O/stem.txt; done
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | cut -c1-160
