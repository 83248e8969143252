# refresh the bench lines that the last changes touch (MX scale rule, xent_sym): default line, G/D-only, C3, C4, C1 + fp8
R=$PWD; O=$R/gpurun_out/prof_r05; mkdir -p $O
export PYTHONPATH=$R
cd $R
timeout 900 python bench.py > $O/r05_bench_c1.json 2> $O/bench.err
timeout 900 python bench.py --pretrained off --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_c1_gd_only.json
timeout 900 python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_c3.json
timeout 900 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_c4_mx_fp8.json
timeout 900 python bench.py --fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 > $O/r05_bench_c1_mx_fp8.json
timeout 900 python tools/bench_conv.py --packed --iters 5 --fp8 2>&1 | grep -v amdgpu > $O/r05_conv_layers_mx_fp8.txt
for f in r05_bench_c1 r05_bench_c1_gd_only r05_bench_c3 r05_bench_c4_mx_fp8 r05_bench_c1_mx_fp8; do tail -1 $O/$f.json | cut -c1-230; done
