#!/bin/bash
# round 6, call m: the older schedule switches re-tested against the new balance of train_g_d (ResNet pullback on the main stream)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_m; mkdir -p $O; cd $R
for rep in 1 2; do
for cfg in "XMC_NOP=1" "XMC_EARLY_ADAM_D=0" "XMC_WGRAD_ASYNC_D=1" "XMC_PREFETCH_EARLY=0" "XMC_HEADS_2STREAM=1" "XMC_WGRAD_ASYNC=1"; do
r=$(env $cfg timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
echo "$cfg -> $r" | tee -a $O/switches.txt
done; done
