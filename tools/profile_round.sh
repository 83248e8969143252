#!/bin/bash
# Regenerates the judged profiling artifacts on the GPU box (run through gpurun from the repo root):
#   profiles/<tag>_rocprofv3_kernel_trace_stats_bench_c1.txt   per-kernel time of `bench.py`
#   profiles/<tag>_pmc_hbm_traffic_per_launch.json              HBM bytes per launch (separate FETCH / WRITE passes)
#   profiles/<tag>_bench_c1.json                                the bench line itself
# usage: bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --graph off --no-instrument > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace/*/*_results.db | head -1) 7 > $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off --no-instrument > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off --no-instrument > $O/write.log 2>&1
python $R/tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_pmc_hbm_traffic_per_launch.json > $O/traffic.txt 2>&1
cd $R && python bench.py > $O/${TAG}_bench_c1.json 2> $O/bench.err
tail -1 $O/${TAG}_bench_c1.json | cut -c1-600
head -30 $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1.txt
head -12 $O/traffic.txt
