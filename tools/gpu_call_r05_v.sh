# fused global-cBN projections + small-kernel latency fixes: tests, then same-box A/Bs
R=$PWD; O=$R/gpurun_out/v; mkdir -p $O
export PYTHONPATH=$R
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_fused_opt.py tests/test_gpu_graph.py tests/test_gpu_kernels.py -x -q -m gpu -k "not resnet" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
B="python bench.py --pretrained off --steps 20 --warmup 3 --no-cpu-baseline --no-gd-only --no-instrument"
bash tools/ab_env.sh XMC_GLOBAL_GB_FUSED 2>&1 | tee $O/ab_global_gb.txt
bash tools/ab_lib.sh xmcgan_image_generation_amd/csrc/build_base/libxmcgan_hip.so 3 $B 2>&1 | cut -c1-200 | tee $O/ab_norm_small_kernels.txt
