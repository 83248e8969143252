#!/bin/bash
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 600 python -m pytest tests -q -m gpu -k "attention or fused_opt or unbiased or independent_flax" -s > $O/tests.log 2>&1
tail -8 $O/tests.log; grep -h "MFMA attention\|fp8 vs bf16\|restored" $O/tests.log | cut -c1-300
for v in 1 0; do echo "XMC_ATTN_MFMA=$v"; XMC_ATTN_MFMA=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'gd_only', d['gd_only']['ms_per_step'])"; done
PH_MASKS="16 32 2" timeout 600 bash tools/phase_abl.sh > $O/phase_abl_staging.txt 2>&1
cat $O/phase_abl_staging.txt
timeout 400 python tools/torch_kernel_sites.py --pretrained off 2>&1 | grep -v amdgpu | head -70 > $O/torch_sites.txt
cat $O/torch_sites.txt
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_graph -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_graph.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace_graph/*/*_results.db | head -1) 7 > $O/r04_trace_mid.txt 2>&1
head -75 $O/r04_trace_mid.txt
rm -rf $O/trace_graph
