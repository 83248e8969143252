#!/bin/bash
# round 5, call H: the whole GPU suite + smoke, then the round's profile set (tools/profile_round.sh r05)
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r05 2>&1 | tail -60
