L=xmcgan_image_generation_amd/libxmcgan_hip.so
cp $L /tmp/full.so
export PYTHONPATH=.
for r in 1 2; do
for n in full 1 2 3 4 8 12 15; do
  if [ $n = full ]; then cp /tmp/full.so $L; else cp libxmcgan_abl_$n.so $L; fi
  echo "abl $n randn: $(python tools/wgrad_time.py 2>/dev/null | tail -1)"
  echo "abl $n zeros: $(python tools/wgrad_time.py zeros 2>/dev/null | tail -1)"
done; done
cp /tmp/full.so $L
