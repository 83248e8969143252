# generator-forward prefetch beside D's optimiser update (HBM-bound beside matrix-bound) vs beside D's backward pass
R=$PWD; O=$R/gpurun_out/w; mkdir -p $O
export PYTHONPATH=$R
bash tools/ab_env.sh XMC_PREFETCH_AT_ADAM 2>&1 | tee $O/ab_prefetch_at_adam.txt
XMC_PREFETCH_AT_ADAM=1 timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_at_adam.txt
