#!/bin/bash
# round 6, call k: same-box A/B of the pointwise kernel's residual prefetch (library built with -DPRE_RES=0 = base)
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_k; mkdir -p $O; cd $R
BASE=xmcgan_image_generation_amd/csrc/build_nopre/libxmcgan_hip.so
timeout 600 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_kernels.py -m gpu -x -q -k "resnet_layer_shape or dual or conv_stream_packed or pw" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
bash tools/ab_lib.sh $BASE 3 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-instrument --no-gd-only | cut -c1-170 | tee $O/ab_step.txt
bash tools/ab_lib.sh $BASE 2 python bench.py --pretrained off --steps 30 --warmup 3 --no-cpu-baseline --no-instrument | cut -c1-170 | tee $O/ab_gd_only.txt
bash tools/ab_lib.sh $BASE 2 python tools/bench_resnet.py | tee $O/ab_resnet.txt
