#!/bin/bash
# round 5, call N: loss assembly in one launch (tests + A/B via git stash is not possible here: bench only); under-filled time of the step
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
export PYTHONPATH=$R
timeout 2400 python -m pytest -q -x -m gpu tests/test_gpu_step.py tests/test_gpu_graph.py tests/test_gpu_dp.py > $O/tests.log 2>&1
tail -4 $O/tests.log
for r in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['gd_only']['ms_per_step'])"; done | tee $O/bench3.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $O/trace_tl -- python $R/bench.py --pretrained off --steps 4 --warmup 2 --no-cpu-baseline --no-gd-only --no-instrument > $O/trace_tl.log 2>&1
DB=$(ls $O/trace_tl/*/*_results.db | head -1)
timeout 300 python $R/tools/underfilled.py $DB 128 40 > $O/r05_underfilled_gd_only.txt 2>&1
timeout 300 python $R/tools/timeline.py $DB > $O/r05_timeline_gd_only.txt 2>&1
rm -rf $O/trace_tl
cat $O/r05_underfilled_gd_only.txt
