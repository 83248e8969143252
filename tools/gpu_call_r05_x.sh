R=$PWD; O=$R/gpurun_out/w; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env.sh XMC_PREFETCH_AT_START 2>&1 | tee $O/ab_prefetch_at_start.txt
