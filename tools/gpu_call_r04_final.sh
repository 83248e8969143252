#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04_bench_c1_final.json 2> $O/bench.err
cut -c1-400 $O/r04_bench_c1_final.json
