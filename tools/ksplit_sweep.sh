# split-K target sweep (XMC_KSPLIT_TARGET = workgroups the split aims at; 640 is the launcher's value) on the small-spatial 3x3 layers
for tag in "D 4 " "D 8     1536" "D 8     768" "G 8  " "G 16    768"; do
  for t in 256 384 512 640 768 1024; do
    echo -n "target $t: "; XMC_KSPLIT_TARGET=$t timeout 300 python tools/bench_conv.py --packed --only "$tag" --iters 40 2>&1 | grep -v "^layer\|TOTAL"
  done
done
