R=$PWD; O=$R/gpurun_out/ac; mkdir -p $O
export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q -m gpu -k "xent or contrastive or step" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --pretrained off --steps 20 --warmup 3 --no-cpu-baseline --no-gd-only --no-instrument"
bash tools/ab_lib.sh xmcgan_image_generation_amd/csrc/build_base/libxmcgan_hip.so 3 $B 2>&1 | cut -c1-200 | tee $O/ab_xent_sym.txt
timeout 600 python tools/torch_kernel_sites.py --pretrained off 2>&1 | grep -v amdgpu > $O/torch_op_sites_gd_only.txt; head -40 $O/torch_op_sites_gd_only.txt | cut -c1-200
