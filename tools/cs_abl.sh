#!/bin/bash
# Compile-time ablation of conv_stream_kernel's k loop and epilogue (conv_stream.hip: CS_ABL) on the C1 layers (per-layer table
# of tools/bench_conv.py --packed, forward + data gradient).  bit 0 no weight refills, bit 1 no patch staging, bit 2 no LDS
# fragment reads, bit 3 no barrier, bit 4 no epilogue.  Run on the GPU box from the repo root.
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/xmcgan_image_generation_amd/libxmcgan_hip.so
C=$R/xmcgan_image_generation_amd/csrc
cp $L /tmp/full.so
export PYTHONPATH=$R
for m in ${CS_MASKS:-1 2 4 8 16 31}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DCS_ABL=$m -c $C/conv_stream.hip -o /tmp/cs_abl_$m.o &
done
wait
for m in full ${CS_MASKS:-1 2 4 8 16 31}; do
  if [ $m = full ]; then cp /tmp/full.so $L; else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/conv_stream\.o$") /tmp/cs_abl_$m.o -o $L; fi
  echo "== CS_ABL $m"
  timeout 300 python $R/tools/bench_conv.py --packed --iters ${CS_ITERS:-5} 2>/dev/null | awk '{print $1,$2,$3,$4,"| fwd",$7,$8,"| dgrad",$10,$11}'
done
cp /tmp/full.so $L
