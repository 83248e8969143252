#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
L=$R/xmcgan_image_generation_amd/libxmcgan_hip.so
C=$R/xmcgan_image_generation_amd/csrc
for f in conv_stream conv_patch conv_stream_mx8; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DXMC_NO_MASK_PRELOAD -c $C/$f.hip -o /tmp/${f}_nopre.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/conv_stream\.o$\|/conv_patch\.o$\|/conv_stream_mx8\.o$") /tmp/conv_stream_nopre.o /tmp/conv_patch_nopre.o /tmp/conv_stream_mx8_nopre.o -o /tmp/lib_nopre.so
cp $L /tmp/lib_pre.so
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mask_bits or conv_phase or conv_stream" 2>&1 | tail -3
for r in 1 2; do for v in nopre pre; do
  cp /tmp/lib_$v.so $L
  echo "== mask words $v (round $r)"
  timeout 200 python tools/bench_mask_bits.py 2>&1 | grep -v amdgpu
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gd_only', d['gd_only']['ms_per_step'])"
done; done
cp /tmp/lib_pre.so $L
