#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 300 python tools/debug_fused_sn.py 2>&1 | grep -v amdgpu | tee $O/debug_fused_sn.txt
for e in "XMC_PREFETCH_G=0" "XMC_OVERLAP_BWD=0" "XMC_OVERLAP_PREP=0"; do
  echo "== fp8 c1 b56 with $e"; env $e timeout 400 python tools/poison_check.py --config c1 --batch 56 --fp8 2>&1 | grep -v amdgpu | grep -E "identical|poison check"
done
timeout 900 bash tools/phase_abl.sh > $O/phase_abl.txt 2>&1
cat $O/phase_abl.txt
timeout 900 python tools/bench_input_pipeline.py --examples 2048 --shards 64 --workers 8 --batches 60 --procs 8,16,32,64 --threads-per-proc 2 2>&1 | grep -v amdgpu | tee $O/pipeline_t2.txt
timeout 600 python tools/bench_input_pipeline.py --examples 2048 --shards 64 --workers 1 --batches 60 --procs 16,32,64 --threads-per-proc 4 2>&1 | grep -v amdgpu | tee $O/pipeline_t4.txt
