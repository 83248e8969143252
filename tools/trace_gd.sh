# per-kernel table of the G/D-only step (serial schedule so that each kernel's time is its own) + stream timeline of the default schedule
R=$PWD; O=$R/gpurun_out/tr_gd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_gd -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_gd.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace_gd/*/*_results.db | head -1) 7 > $O/stats_gd.txt 2>&1
timeout 900 rocprofv3 --kernel-trace -d $O/trace_tl -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_tl.log 2>&1
timeout 900 python $R/tools/timeline.py $(ls $O/trace_tl/*/*_results.db | head -1) > $O/timeline_gd.txt 2>&1
rm -rf $O/trace_gd $O/trace_tl
head -40 $O/timeline_gd.txt
