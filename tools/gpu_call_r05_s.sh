# is the neighbour penalty cache pollution (its 256 MiB cyclic buffer = the Infinity Cache's size) or occupancy?
R=$PWD; O=$R/gpurun_out/cont; mkdir -p $O
export PYTHONPATH=$R
for mib in 256 32 2; do
  timeout 600 python tools/cu_contention.py --ks 0,1,8 --pretrained off --buf-mib $mib > $O/cont_buf$mib.txt 2>&1
done
timeout 600 python tools/cu_contention.py --ks 0,1,8,32 --pretrained off --kind mfma > $O/cont_mfma.txt 2>&1
for f in $O/cont_buf256.txt $O/cont_buf32.txt $O/cont_buf2.txt $O/cont_mfma.txt; do echo == $f; tail -n 6 $f | cut -c1-160; done
