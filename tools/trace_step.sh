R=$PWD; O=$R/gpurun_out/tr_phase; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace/*/*_results.db | head -1) 7 > $O/stats.txt 2>&1
head -45 $O/stats.txt | cut -c1-150
