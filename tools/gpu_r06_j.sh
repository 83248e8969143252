#!/bin/bash
# round 6, call j: the whole GPU suite on the final code + smoke + the default bench line
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_j; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -22 $O/pytest.log | cut -c1-160
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_c1.json 2> $O/bench.err; tail -1 $O/bench_c1.json | cut -c1-330
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29590 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | cut -c1-200 | tee $O/bench_torchrun_world1.json
