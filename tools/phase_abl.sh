#!/bin/bash
# Compile-time ablation of the phase kernels' k loop (conv_stream.hip: PH_ABL): where do conv_phase4_kernel / conv_phase_kernel
# spend their time?  Builds one library per ablation mask next to the product library and times the 10 resampling-adjacent C1
# layers with each (forward + data gradient, phase launches only).  Run on the GPU box from the repo root.
#   bit 0 no weight refills, bit 1 no patch staging, bit 2 no LDS fragment reads, bit 3 no barrier
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/xmcgan_image_generation_amd/libxmcgan_hip.so
C=$R/xmcgan_image_generation_amd/csrc
cp $L /tmp/full.so
export PYTHONPATH=$R
for m in ${PH_MASKS:-1 2 4 8 15}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DPH_ABL=$m -c $C/conv_stream.hip -o /tmp/cs_abl_$m.o &
done
wait
for r in ${PH_ROUNDS:-1}; do
for m in full ${PH_MASKS:-1 2 4 8 15}; do
  if [ $m = full ]; then cp /tmp/full.so $L; else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/conv_stream\.o$") /tmp/cs_abl_$m.o -o $L; fi
  echo "== PH_ABL $m (round $r)"
  timeout 200 python $R/tools/bench_phase.py --only-phase --iters 3 2>/dev/null | awk '{print $1,$2,$3,"| fwd",$9,$10,"| dgrad",$17,$18}'
done; done
cp /tmp/full.so $L
