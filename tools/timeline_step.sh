# graph-replay timeline of the default workload: kernels in flight, what runs alone (tools/timeline.py)
R=$PWD; O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-gd-only --no-instrument ${1:-} > $O/trace.log 2>&1
python $R/tools/timeline.py $(ls $O/trace/*/*_results.db | head -1) > $O/timeline.txt 2>&1
head -60 $O/timeline.txt
