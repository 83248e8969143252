# N launches of the world-1 torchrun bench (hipGraph with the RCCL calls inside): does every launch survive its capture?
# usage: bash tools/torchrun_repeat.sh [N] [env assignments ...]
N=${1:-6}; shift
O=gpurun_out/torchrun_repeat; mkdir -p $O
ok=0
for i in $(seq 1 $N); do
  env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>$O/err_$i.txt | tail -1 > $O/out_$i.json
  if python -c "import json,sys; d=json.loads(open('$O/out_$i.json').read()); print('launch $i:', d['ms_per_step'], d['launch_mode'])" 2>/dev/null; then ok=$((ok+1)); else echo "launch $i FAILED: $(grep -m1 -i 'error\|abort' $O/err_$i.txt | cut -c1-200)"; fi
done
echo "$ok of $N launches completed"
