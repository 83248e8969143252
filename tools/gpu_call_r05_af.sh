# runtime environment knobs (not library code): do they move the replayed step?
R=$PWD; O=$R/gpurun_out/af; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env_values.sh HIP_FORCE_DEV_KERNARG 0 1 2>&1 | cut -c1-70 | tee -a $O/ab_runtime_env.txt
bash tools/ab_env_values.sh GPU_MAX_HW_QUEUES 4 2 8 2>&1 | cut -c1-70 | tee -a $O/ab_runtime_env.txt
bash tools/ab_env_values.sh HSA_NO_SCRATCH_RECLAIM 0 1 2>&1 | cut -c1-70 | tee -a $O/ab_runtime_env.txt
