#!/bin/bash
# round 5, call F: bf16 gamma / beta maps of the local cBN sites -- kernel + step tests, same-box A/B; kernel times of the fused
# optimiser + preparation kernel vs the separate passes (serial schedule)
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "cbn" > $O/tests_cbn.log 2>&1
tail -3 $O/tests_cbn.log
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_gpu_fused_opt.py > $O/tests.log 2>&1
tail -8 $O/tests.log
bash tools/ab_env.sh XMC_GB_BF16 2>&1 | tee $O/ab_gb_bf16.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  XMC_FUSE_PREP=$v XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_fp$v -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_fp$v.log 2>&1
  timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace_fp$v/*/*_results.db | head -1) 7 > $O/stats_fuse_prep_$v.txt 2>&1
  rm -rf $O/trace_fp$v
  grep -E "adam|wprep|sn_|TOTAL" $O/stats_fuse_prep_$v.txt
done
