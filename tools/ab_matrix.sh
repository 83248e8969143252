#!/bin/bash
# A/B of the stream-level schedule switches (run on the GPU box): ms/step and host ms of one step, hipGraph replay and eager
for d in 0 1; do for o in 0 1; do for g in on off; do
r=$(XMC_WGRAD_ASYNC_D=$d XMC_OVERLAP_BWD=$o timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument --graph $g 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])")
echo "async_wgrad_in_train_d=$d overlap_pullbacks=$o graph=$g -> $r"
done; done; done
