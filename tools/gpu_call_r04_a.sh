#!/bin/bash
# round 4, GPU call A: the new evidence-gap tests, the MFMA-busy PMC pass, the repaired MX-fp8 per-layer table, a bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_mx8.py tests/test_gpu_step.py -x -q -m gpu -k "c4_workload or unbiased or independent_flax" -s > $O/new_tests.log 2>&1
tail -5 $O/new_tests.log
grep -h "fp8 vs bf16\|C4 \|restored" $O/new_tests.log | cut -c1-400
timeout 600 python tools/bench_conv.py --packed --iters 5 --fp8 2>&1 | grep -v amdgpu > $O/r04_conv_layers_mx_fp8.txt
tail -3 $O/r04_conv_layers_mx_fp8.txt
cd /tmp && export TMPDIR=/tmp
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq -o s --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/sq.log 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o m --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/mfma.log 2>&1
tail -2 $O/mfma.log | cut -c1-300
cd $R
python tools/pmc_sq.py $O/sq $O/mfma > $O/r04_pmc_sq_per_kernel_baseline.txt 2>$O/pmc_sq.err
head -25 $O/r04_pmc_sq_per_kernel_baseline.txt
ls $O/mfma/* | head; head -3 $(ls $O/mfma/*/*counter_collection.csv | head -1)
rm -rf $O/sq $O/mfma/*/*.db
python bench.py --steps 20 --warmup 5 > $O/r04_bench_c1_baseline.json 2> $O/bench.err
cut -c1-700 $O/r04_bench_c1_baseline.json
