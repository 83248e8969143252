R=$PWD; O=$R/gpurun_out/ak; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env_values.sh DEBUG_CLR_GRAPH_PACKET_CAPTURE 1 0 2>&1 | cut -c1-80 | tee -a $O/ab_runtime_env2.txt
bash tools/ab_env_values.sh HSA_ENABLE_INTERRUPT 1 0 2>&1 | cut -c1-80 | tee -a $O/ab_runtime_env2.txt
bash tools/ab_env_values.sh HIP_LAUNCH_BLOCKING 0 2>&1 | cut -c1-80 | tee -a $O/ab_runtime_env2.txt
