R=$PWD; O=$R/gpurun_out/tr_now; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_graph -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_graph.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace_graph/*/*_results.db | head -1) 7 > $O/stats_c1.txt 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_gd -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_gd.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace_gd/*/*_results.db | head -1) 7 > $O/stats_gd.txt 2>&1
rm -rf $O/trace_graph $O/trace_gd
cd $R; PYTHONPATH=. timeout 300 python tools/torch_kernel_sites.py --pretrained on > $O/sites_on.txt 2>&1
head -5 $O/stats_c1.txt
