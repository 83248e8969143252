#!/bin/bash
# Builds libxmcgan_pwabl_<bits>.so (repo root) = the library with conv_stream.hip compiled at -DPW_ABL=<bits>; run
# tools/pw_abl.sh on the GPU box afterwards.  The other objects come from the normal build (make first).
set -e
cd "$(dirname "$0")/../xmcgan_image_generation_amd/csrc"
make >/dev/null
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DPW_ABL=$b -c conv_stream.hip -o build/conv_stream_pwabl_$b.o &
done
wait
for b in "$@"; do
  objs=$(ls build/*.o | grep -v "conv_stream.o\|pwabl")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv_stream_pwabl_$b.o -o ../../libxmcgan_pwabl_$b.so
done
ls -la ../../libxmcgan_pwabl_*.so
