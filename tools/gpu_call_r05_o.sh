#!/bin/bash
# round 5, call O: two executable graphs replayed in turn (no host-side wait on the previous replay of the same exec)
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_graph.py tests/test_gpu_step.py -k "graph or benchmarked_workload_c1" > $O/tests.log 2>&1
tail -4 $O/tests.log
bash tools/ab_env_values.sh XMC_GRAPH_EXECS 1 2 3 2>&1 | tee $O/ab_graph_execs.txt
