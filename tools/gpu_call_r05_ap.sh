R=$PWD; O=$R/gpurun_out/ap; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/tr -- python $R/bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/tr.log 2>&1
timeout 300 python $R/tools/rocpd_stats.py $(ls $O/tr/*/*_results.db | head -1) 7 > $O/c4_kernel_stats.txt 2>&1
rm -rf $O/tr
grep -i "mx8\|TOTAL" $O/c4_kernel_stats.txt | cut -c1-140; head -14 $O/c4_kernel_stats.txt | cut -c1-140
