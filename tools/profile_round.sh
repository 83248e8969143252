#!/bin/bash
# Regenerates the judged profiling artifacts on the GPU box (run through gpurun from the repo root):
#   profiles/<tag>_rocprofv3_kernel_trace_stats_bench_c1.txt   per-kernel time of `bench.py`
#   profiles/<tag>_pmc_hbm_traffic_per_launch.json              HBM bytes per launch (separate FETCH / WRITE passes)
#   profiles/<tag>_bench_c1.json                                the bench line itself
# usage: bash tools/profile_round.sh r01
set -u
exec < /dev/null      # nothing here may ever wait on a terminal (a stray `head` without a file cost 40 GPU-minutes in round 4)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# serial schedule (one stream: every kernel alone on the GPU).  Primary trace: the step REPLAYED as a hipGraph -- kernels back
# to back, the chip at the clock it holds inside a real step, the condition bench.py's instrumented step reproduces with its
# sleeping wave; second trace: eager launches (the host is slower than the GPU: gaps between kernels, a cooler chip).
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_graph -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_graph.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace_graph/*/*_results.db | head -1) 7 > $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1.txt 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/trace.log 2>&1
timeout 900 python $R/tools/rocpd_stats.py $(ls $O/trace/*/*_results.db | head -1) 7 > $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1_eager_launches.txt 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/fetch.log 2>&1
XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/write.log 2>&1
timeout 900 python $R/tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_pmc_hbm_traffic_per_launch.json > $O/traffic.txt 2>&1
cd $R && timeout 900 python bench.py > $O/${TAG}_bench_c1.json 2> $O/bench.err
tail -1 $O/${TAG}_bench_c1.json | cut -c1-600
head -30 $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1.txt
head -12 $O/traffic.txt
# (round 4: the unchanged round-3 probes -- wgrad split sweep, data-power, load-path, few-pixel pointwise, HBM copy rate, RCCL loopback --
#  are not re-run; their r03 files stay the reference)
# extra round artefacts: per-layer convolution table, schedule A/B matrix, SQ counter pass, C3 and batch-2 bench lines
cd $R
timeout 900 python tools/bench_conv.py --packed 2>&1 | grep -v amdgpu > $O/${TAG}_conv_layers_bf16.txt
timeout 900 python tools/bench_conv.py --packed --iters 5 --fp8 2>&1 | grep -v amdgpu > $O/${TAG}_conv_layers_mx_fp8.txt
timeout 900 python bench.py --config c4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_c4_mx_fp8.json
timeout 900 python bench.py --fp8 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 > $O/${TAG}_bench_c1_mx_fp8.json
PYTHONPATH=$R timeout 900 python tools/bench_phase.py 2>&1 | grep -v amdgpu > $O/${TAG}_conv_phase_vs_3x3.txt
PYTHONPATH=$R timeout 900 python tools/mfma_rate_probe.py 2>&1 | grep -v amdgpu > $O/${TAG}_mfma_rate_probe.txt
XMC_PHASE_CONV=0 timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_c1_phase_conv_off.json
timeout 900 python tools/bench_gemm.py 2>&1 | grep -v amdgpu > $O/${TAG}_gemm_word_loss_shapes.txt
timeout 900 python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_c3.json
timeout 900 python bench.py --batch 2 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 > $O/${TAG}_bench_c1_batch2.json
timeout 900 python bench.py --graph off --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 > $O/${TAG}_bench_c1_eager.json
timeout 900 python bench.py --pretrained off --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_c1_gd_only.json
XMC_DP_OVERLAP=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>$O/torchrun1.err | tail -1 > $O/${TAG}_bench_c1_torchrun_world1_graph_program_order.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>$O/torchrun2.err | tail -1 > $O/${TAG}_bench_c1_torchrun_world1_graph_dp_overlap.json
timeout 900 python tools/bench_resnet.py --detail 2>&1 | grep -v amdgpu > $O/${TAG}_resnet50_path_per_launch.txt
(cd /tmp && XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq -o s --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > /dev/null 2>&1)
# matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA is executing, summed over the SIMDs) against GRBM_GUI_ACTIVE
# (chip-busy cycles of the dispatch) -- busy % of the 1024 SIMDs' matrix pipes and, with the kernel's wall time, the clock it held
(cd /tmp && XMC_OVERLAP_BWD=0 XMC_PREFETCH_G=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o m --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gd-only --graph off --no-instrument > $O/mfma.log 2>&1)
timeout 900 python tools/pmc_sq.py $O/sq $O/mfma > $O/${TAG}_pmc_sq_per_kernel.txt 2>$O/pmc_sq.err
PYTHONPATH=$R timeout 400 python tools/bench_floor.py 2>&1 | grep -v amdgpu > $O/${TAG}_floor_pieces.txt
PYTHONPATH=$R timeout 600 python tools/cu_contention.py 2>&1 | grep -v amdgpu > $O/${TAG}_cu_contention.txt
timeout 900 python tools/bench_input_pipeline.py --examples 1536 --shards 48 --workers 1,8 --batches 40 --procs 8,12,14,15,16 --threads-per-proc 1,2 2>&1 | grep -v "amdgpu\|resource_tracker\|warnings.warn" > $O/${TAG}_input_pipeline_decode_rate.txt
XMC_DP_BUCKET_D=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>$O/torchrun3.err | tail -1 > $O/${TAG}_bench_c1_torchrun_world1_graph_d_exchange_in_one_piece.json
cut -c1-200 $O/${TAG}_bench_c1_torchrun_world1_graph_program_order.json; cut -c1-200 $O/${TAG}_bench_c1_torchrun_world1_graph_dp_overlap.json; cut -c1-200 $O/${TAG}_bench_c4_mx_fp8.json; cut -c1-200 $O/${TAG}_bench_c3.json; cut -c1-200 $O/${TAG}_bench_c1_batch2.json; cut -c1-200 $O/${TAG}_bench_c1_eager.json; cut -c1-200 $O/${TAG}_bench_c1_gd_only.json
# ---- round 6 additions
# (a) the exclusive exchange schedule at world 1 (input of tools/model_scaling.py's second row) and the modelled-scaling table
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29564 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument --grad-schedule exclusive 2>$O/torchrun4.err | tail -1 > $O/${TAG}_bench_c1_torchrun_world1_graph_dp_exclusive.json
# (b) RCCL channel count inside the graph at world 1: what the collective's kernels cost the step when they move nothing
for ch in 1 2 4 8 16 32; do
  echo -n "NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1: " >> $O/${TAG}_rccl_channels_world1.txt
  NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29570 + ch)) bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument --grad-schedule overlapped 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step')" >> $O/${TAG}_rccl_channels_world1.txt 2>&1
done
# (c) per-family kernel time inside the replayed graph of the DEFAULT (overlapped) schedule -> bench.py's roofline.in_replayed_graph
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_graph_default -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_graph_default.log 2>&1)
timeout 300 python $R/tools/graph_family_time.py $(ls $O/trace_graph_default/*/*_results.db | head -1) 7 $O/${TAG}_kernel_time_in_replayed_graph.json > /dev/null 2>$O/graph_family.err
timeout 300 python $R/tools/rocpd_stats.py $(ls $O/trace_graph_default/*/*_results.db | head -1) 7 > $O/${TAG}_rocprofv3_kernel_trace_stats_bench_c1_default_schedule.txt 2>&1
cat $O/${TAG}_rccl_channels_world1.txt; head -c 600 $O/${TAG}_kernel_time_in_replayed_graph.json
