#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
export PYTHONPATH=$R
L=$R/xmcgan_image_generation_amd/libxmcgan_hip.so
C=$R/xmcgan_image_generation_amd/csrc
# library variant: weight ring depth 2 in the phases-as-waves kernels (the round-3 depth)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DPH4_DEPTH=2 -c $C/conv_stream.hip -o /tmp/cs_d2.o &
timeout 1700 python -m pytest tests -q -m gpu --maxfail=30 > $O/gpu_tests.log 2>&1
tail -30 $O/gpu_tests.log
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/conv_stream\.o$") /tmp/cs_d2.o -o /tmp/lib_d2.so
cp $L /tmp/lib_d4.so
for r in 1 2; do for v in d2 d4; do
  cp /tmp/lib_$v.so $L
  echo "== ring depth $v (round $r)"
  timeout 200 python tools/bench_phase.py --only-phase --iters 3 2>/dev/null | awk '{print $1,$2,$3,"| fwd",$9,$10,"| dgrad",$18,$19}'
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gd_only', d['gd_only']['ms_per_step'])"
done; done
cp /tmp/lib_d4.so $L
echo "== fp8 (serial schedule by default now)"
timeout 400 python tools/poison_check.py --config c1 --batch 56 --fp8 2>&1 | grep -v amdgpu | grep -E "identical|poison check"
for p in "XMC_PREFETCH_G=0 XMC_OVERLAP_BWD=0" "XMC_PREFETCH_G=0 XMC_OVERLAP_PREP=0" "XMC_OVERLAP_BWD=0 XMC_OVERLAP_PREP=0"; do
  echo "== fp8 overlapped except: $p"; env XMC_FP8_OVERLAP=1 $p timeout 400 python tools/poison_check.py --config c1 --batch 56 --fp8 2>&1 | grep -v amdgpu | grep -E "identical|poison check"
done
timeout 300 python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | cut -c1-250
timeout 300 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | cut -c1-250
XMC_FP8_OVERLAP=1 timeout 300 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | cut -c1-250
timeout 900 python tools/bench_input_pipeline.py --examples 1536 --shards 48 --workers 1,16 --batches 40 --procs 8,12,16 --threads-per-proc 1 2>&1 | grep -v "amdgpu\|resource_tracker\|warnings.warn" | tee $O/pipeline_c_decode_t1.txt
timeout 600 python tools/bench_input_pipeline.py --examples 1536 --shards 48 --workers 8 --batches 40 --procs 6,8,12 --threads-per-proc 2 2>&1 | grep -v "amdgpu\|resource_tracker\|warnings.warn" | tee $O/pipeline_c_decode_t2.txt
