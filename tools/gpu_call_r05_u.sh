# input pipeline decode rate on the box's host cores (no GPU work)
R=$PWD; O=$R/gpurun_out/pipe; mkdir -p $O
export PYTHONPATH=$R
timeout 1500 python tools/bench_input_pipeline.py --examples 1536 --shards 48 --workers 1,8 --batches 40 --procs 8,12,14,15,16 --threads-per-proc 1,2 2>&1 | grep -v "amdgpu\|resource_tracker\|warnings.warn" > $O/r05_input_pipeline_decode_rate.txt
cat $O/r05_input_pipeline_decode_rate.txt
