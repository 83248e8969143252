# launch-heuristic knobs re-swept on the final code (one box)
R=$PWD; O=$R/gpurun_out/ai; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env_values.sh XMC_TILE64_PCT 100 60 150 2>&1 | cut -c1-64 | tee -a $O/sweeps_final_code.txt
bash tools/ab_env_values.sh XMC_KSPLIT_TARGET 256 192 320 2>&1 | cut -c1-64 | tee -a $O/sweeps_final_code.txt
bash tools/ab_env_values.sh XMC_KSPLIT_TARGET_PW 256 128 384 2>&1 | cut -c1-64 | tee -a $O/sweeps_final_code.txt
