for rep in 1 2; do
for cfg in "XMC_WGRAD_ASYNC_D=0 XMC_WGRAD_ASYNC=0" "XMC_WGRAD_ASYNC_D=1 XMC_WGRAD_ASYNC=0" "XMC_WGRAD_ASYNC_D=0 XMC_PREFETCH_G=0" "XMC_OVERLAP_PREP=0"; do
r=$(env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
echo "$cfg -> $r"
done; done
