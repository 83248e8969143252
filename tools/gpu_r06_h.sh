#!/bin/bash
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/timeline_step.sh > gpurun_out/r06_h_timeline.txt 2>&1
cp gpurun_out/tl/timeline.txt gpurun_out/r06_h_timeline_full.txt 2>/dev/null
head -70 gpurun_out/r06_h_timeline.txt
