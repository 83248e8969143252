#!/bin/bash
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -25 $O/pytest.log | cut -c1-250
