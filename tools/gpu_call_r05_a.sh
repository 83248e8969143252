#!/bin/bash
# round 5, call A: the new full-size parity cases + the D-block change against the oracle; A/B of relu(x) from the pooling pass
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_word_loss_fused.py "tests/test_gpu_kernels.py::test_attention_for_g_on_mfma" \
  "tests/test_gpu_kernels.py::test_pointwise" tests/test_gpu_step.py::test_train_step_fp32_c1_shapes_batch8 \
  tests/test_gpu_step.py::test_train_step_bf16_c1_shapes_batch8_vs_oracle tests/test_gpu_step.py::test_train_step_fp32_tiny \
  tests/test_gpu_step.py::test_benchmarked_workload_product_optimiser_mode_equals_the_test_sessions \
  tests/test_gpu_step.py::test_train_step_is_bit_reproducible tests/test_gpu_graph.py > $O/tests.log 2>&1
tail -8 $O/tests.log
cat gpurun_out/parity_measured.txt 2>/dev/null
bash tools/ab_env.sh XMC_RELU_X 2>&1 | tee $O/ab_relu_x.txt
