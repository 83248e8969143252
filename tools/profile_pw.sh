R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pw; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for args in "--cin 64 --cout 256 --res" "--cin 256 --cout 64" "--h 16 --cin 1024 --cout 256" "--h 16 --cin 256 --cout 1024 --res" "--h 8 --cin 2048 --cout 512"; do
  python $R/tools/bench_pw.py $args 2>&1 | grep -v amdgpu; python $R/tools/bench_pw.py $args --plain 2>&1 | grep -v amdgpu | sed 's/^/   plain: /'
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/tools/bench_pw.py --cin 64 --cout 256 --res --iters 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/tools/bench_pw.py --cin 64 --cout 256 --res --iters 5 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $O/fetch $O/write | grep -i "conv_pw"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq -o s --output-format csv -- python $R/tools/bench_pw.py --cin 64 --cout 256 --res --iters 5 > /dev/null 2>&1
python $R/tools/pmc_sq.py $O/sq 2>/dev/null | grep -iE "kernel|conv_pw"
