# does the neighbour's L2 footprint carry the penalty?  same stream, four cache policies
R=$PWD; O=$R/gpurun_out/cont; mkdir -p $O
export PYTHONPATH=$R
for pol in 0 1 2 3; do
  timeout 600 python tools/cu_contention.py --ks 0,1,8 --pretrained off --policy $pol > $O/cont_policy$pol.txt 2>&1
done
for pol in 0 1 2 3; do echo == policy $pol; tail -n 6 $O/cont_policy$pol.txt | cut -c1-160; done
