#!/bin/bash
# Build a VARIANT of libxmcgan_hip.so with extra compiler flags into csrc/build_<tag>/ (for tools/ab_lib.sh same-box A/Bs).
# usage: bash tools/build_variant.sh <tag> <flags...>    ->  xmcgan_image_generation_amd/csrc/build_<tag>/libxmcgan_hip.so
set -e
TAG=$1; shift
cd "$(dirname "$0")/../xmcgan_image_generation_amd/csrc"
mkdir -p build_$TAG
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
for f in $SRCS; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c $f -o build_$TAG/${f%.hip}.o ) &
  if (( $(jobs -r | wc -l) >= 8 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_$TAG/*.o -o build_$TAG/libxmcgan_hip.so
ls -la build_$TAG/libxmcgan_hip.so
