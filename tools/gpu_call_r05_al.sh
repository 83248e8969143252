# whole-library compiler scheduling strategies (variant builds) against the shipped build, one box
R=$PWD; O=$R/gpurun_out/al; mkdir -p $O
export PYTHONPATH=$R
B="python bench.py --pretrained off --steps 20 --warmup 3 --no-cpu-baseline --no-gd-only --no-instrument"
L=xmcgan_image_generation_amd/libxmcgan_hip.so
cp $L /tmp/shipped.so
for r in 1 2; do
  for v in shipped ilp mclause; do
    if [ $v = shipped ]; then cp /tmp/shipped.so $L; else cp xmcgan_image_generation_amd/csrc/build_$v/libxmcgan_hip.so $L; fi
    echo "$v r$r: $($B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v,3) for k,v in d['losses'].items()})")"
  done
done 2>&1 | tee $O/ab_sched_strategy.txt
cp /tmp/shipped.so $L
