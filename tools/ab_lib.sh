#!/bin/bash
# Same-box A/B of two builds of libxmcgan_hip.so: alternates the library under the package between runs of a command.
# usage (through gpurun, repo root): bash tools/ab_lib.sh <base.so> <rounds> <command...>   -- prints the command's last line per run
BASE=$1; ROUNDS=$2; shift 2
L=xmcgan_image_generation_amd/libxmcgan_hip.so
cp $L /tmp/ab_new.so
for r in $(seq $ROUNDS); do
  cp $BASE $L;          echo "base r$r: $("$@" 2>/dev/null | tail -1 | cut -c1-400)"
  cp /tmp/ab_new.so $L; echo "new  r$r: $("$@" 2>/dev/null | tail -1 | cut -c1-400)"
done
cp /tmp/ab_new.so $L
