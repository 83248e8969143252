# schedule switches re-tested on the final code (the round's lesson: an HBM-bound kernel beside matrix-bound ones is not free)
R=$PWD; O=$R/gpurun_out/ae; mkdir -p $O
export PYTHONPATH=$R
for sw in XMC_OVERLAP_PREP XMC_EARLY_ADAM_D XMC_WGRAD_ASYNC_D; do bash tools/ab_env.sh $sw 2>&1 | cut -c1-80 | tee -a $O/ab_schedule_switches.txt; done
