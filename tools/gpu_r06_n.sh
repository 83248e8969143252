#!/bin/bash
# round 6, call n: 64-cout tiles for more of the few-tile 3x3 launches (tile64_pct 100 -> 150 / 200 / 300)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_n; mkdir -p $O; cd $R
for rep in 1 2; do
for v in 100 150 200 300; do
echo -n "XMC_TILE64_PCT=$v resnet path: " | tee -a $O/tile64.txt
XMC_TILE64_PCT=$v timeout 200 python tools/bench_resnet.py 2>&1 | grep TOTAL | tee -a $O/tile64.txt
r=$(XMC_TILE64_PCT=$v timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
echo "XMC_TILE64_PCT=$v step -> $r" | tee -a $O/tile64.txt
done; done
