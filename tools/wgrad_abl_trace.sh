L=xmcgan_image_generation_amd/libxmcgan_hip.so
cp $L /tmp/full.so
export PYTHONPATH=$PWD
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in full 64 128 4; do
  if [ $n = full ]; then cp /tmp/full.so $R/$L; else cp $R/libxmcgan_abl_$n.so $R/$L; fi
  rocprofv3 --kernel-trace --stats -d /tmp/tr_$n -- python $R/tools/wgrad_time.py > /dev/null 2>&1
  echo "== $n"; python $R/tools/rocpd_stats.py $(ls /tmp/tr_$n/*/*_results.db | head -1) 7 2>&1 | head -14 | cut -c1-220
done
cp /tmp/full.so $R/$L
