# which kernels pay for a resident neighbour; does sizing the launch targets below one full round remove it?
R=$PWD; O=$R/gpurun_out/cont; mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 rocprofv3 --kernel-trace -d $O/tr$k -- python $R/tools/cu_contention.py --ks $k --steps 4 --pretrained off > $O/tr$k.log 2>&1
done
timeout 300 python $R/tools/contention_kernels.py $(ls $O/tr0/*/*_results.db | head -1) $(ls $O/tr1/*/*_results.db | head -1) > $O/contention_kernels.txt 2>&1
rm -rf $O/tr0 $O/tr1
cd $R
timeout 600 python tools/cu_contention.py --ks 0,1,8 --pretrained off > $O/cont_default.txt 2>&1
XMC_KSPLIT_TARGET=224 XMC_KSPLIT_TARGET_PHASE=336 XMC_KSPLIT_TARGET_PW=224 XMC_WGRAD_TARGET_HI=336 XMC_WGRAD_TARGET_LO=448 XMC_WGRAD_TARGET_PHASE=336 \
  timeout 600 python tools/cu_contention.py --ks 0,1,8 --pretrained off > $O/cont_targets_7of8.txt 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 python tools/cu_contention.py --ks 0,1,8 --pretrained off > $O/cont_serial.txt 2>&1
head -70 $O/contention_kernels.txt; tail -4 $O/cont_default.txt $O/cont_targets_7of8.txt $O/cont_serial.txt
