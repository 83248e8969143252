R=$PWD; O=$R/gpurun_out/af; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env_values.sh GPU_MAX_HW_QUEUES 4 3 5 6 2>&1 | cut -c1-70 | tee -a $O/ab_hw_queues.txt
GPU_MAX_HW_QUEUES=2 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-instrument --no-gd-only 2>&1 | tail -5 | cut -c1-300
