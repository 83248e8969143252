#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 400 python tools/bench_floor.py 2>&1 | grep -v amdgpu | tee $O/r04_floor_pieces.txt
timeout 600 python tools/cu_contention.py 2>&1 | grep -v amdgpu | tee $O/r04_cu_contention.txt
