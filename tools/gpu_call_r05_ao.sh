R=$PWD; O=$R/gpurun_out/an; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python tools/fp8_logit_check.py --init 2 2>&1 | grep -v "amdgpu\|Warning\|warn" | tee $O/fp8_logits_init2.txt
