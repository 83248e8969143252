#!/bin/bash
# round 6, call i: the judged profiling artefacts of the round (tools/profile_round.sh r06) + the modelled-scaling table
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/profile_round.sh r06 > gpurun_out/r06_i_profile_round.log 2>&1
O=$R/gpurun_out/prof_r06
mkdir -p $R/profiles_tmp && cp $O/r06_* $R/profiles_tmp/ 2>/dev/null
# model_scaling reads profiles/<tag>_*: point it at the fresh files
for f in r06_bench_c1.json r06_bench_c1_torchrun_world1_graph_dp_overlap.json r06_bench_c1_torchrun_world1_graph_dp_exclusive.json r06_cu_contention.txt; do cp $O/$f $R/profiles/ 2>/dev/null; done
python tools/model_scaling.py r06 > $O/r06_modelled_scaling.txt 2>&1
tail -12 $O/r06_modelled_scaling.txt
tail -5 gpurun_out/r06_i_profile_round.log | cut -c1-300
ls $O | head -60
rm -rf $R/profiles_tmp
