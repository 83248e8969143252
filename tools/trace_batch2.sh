# per-kernel table of the B = 2 step (the batch-independent floor): serial schedule so every kernel's time is its own
R=$PWD; O=$R/gpurun_out/tr_b2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --batch 2 --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace/*/*_results.db | head -1) 7 > $O/stats_b2.txt 2>&1
rm -rf $O/trace
head -70 $O/stats_b2.txt
