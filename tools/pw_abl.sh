#!/bin/bash
# On the GPU box: tools/pw_small_m.py with each ablated library of tools/pw_abl_build.sh in place of the real one.
L=xmcgan_image_generation_amd/libxmcgan_hip.so
cp $L /tmp/full.so
export PYTHONPATH=.
for n in full "$@"; do
  if [ $n = full ]; then cp /tmp/full.so $L; else cp libxmcgan_pwabl_$n.so $L; fi
  echo "== PW_ABL $n"; python tools/pw_small_m.py --variants 1 2>/dev/null | grep -v amdgpu
done
cp /tmp/full.so $L
