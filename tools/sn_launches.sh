R=$PWD; O=$R/gpurun_out/sn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 rocprofv3 --kernel-trace -d $O/trace -- python $R/bench.py --pretrained off --steps 3 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/log 2>&1
DB=$(ls $O/trace/*/*_results.db | head -1)
python $R/tools/kernel_launches.py $DB sn_matvec 24
python $R/tools/kernel_launches.py $DB sn_colsum 6
python $R/tools/kernel_launches.py $DB wprep_kernel 6
python $R/tools/kernel_launches.py $DB sn_dot_kernel 6
rm -rf $O/trace
