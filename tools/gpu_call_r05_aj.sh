R=$PWD; O=$R/gpurun_out/aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/tr -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/tr.log 2>&1
timeout 300 python $R/tools/torch_kernels_in_trace.py $(ls $O/tr/*/*_results.db | head -1) 7 > $O/torch_kernels_gd_only.txt 2>&1
rm -rf $O/tr
cat $O/torch_kernels_gd_only.txt | cut -c1-130
