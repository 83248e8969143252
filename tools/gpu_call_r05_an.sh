R=$PWD; O=$R/gpurun_out/an; mkdir -p $O
export PYTHONPATH=$R
timeout 1500 python tools/fp8_bias_over_inits.py --inits 8 2>&1 | grep -v "amdgpu\|Warning\|warn" | tee $O/fp8_bias_over_inits.txt
