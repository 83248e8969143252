# D's optimiser update in slices from inside its backward pass (train_d): tests, then same-box A/B
R=$PWD; O=$R/gpurun_out/ad; mkdir -p $O
export PYTHONPATH=$R
timeout 1800 python -m pytest tests/test_gpu_fused_opt.py tests/test_gpu_graph.py tests/test_gpu_step.py tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
bash tools/ab_env.sh XMC_ADAM_SLICES 2>&1 | tee $O/ab_adam_slices.txt
