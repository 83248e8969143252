for m in wgrad mix; do PYTHONPATH=. timeout 300 python tools/mx8_concurrency2.py $m 2>&1 | grep -v "amdgpu\|^   " | cut -c1-110; done
B="XMC_FP8_OVERLAP=1"
for sched in "XMC_PREFETCH_G=1 XMC_OVERLAP_BWD=0 XMC_OVERLAP_PREP=0" "XMC_PREFETCH_G=0 XMC_OVERLAP_BWD=1 XMC_OVERLAP_PREP=0" "XMC_PREFETCH_G=1 XMC_OVERLAP_BWD=1 XMC_OVERLAP_PREP=1"; do
  env $B $sched timeout 400 python tools/fp8_race_hunt.py --runs 4 2>&1 | grep -v amdgpu | head -6
done
env $B timeout 400 python tools/fp8_race_hunt.py --config c1 --batch 56 --runs 3 2>&1 | grep -v amdgpu | head -6
