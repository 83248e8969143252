#!/bin/bash
# round 4, GPU call B: full GPU suite on the folded / fused path, step A/Bs of the new switches, phase-kernel ablations
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 -x -k "fused_opt" -s > $O/fused_tests.log 2>&1
tail -15 $O/fused_tests.log
timeout 1500 python -m pytest tests -q -m gpu --maxfail=12 > $O/gpu_tests.log 2>&1
tail -25 $O/gpu_tests.log
B="timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-instrument"
for r in 1; do
  $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default            ', d['ms_per_step'], d['gd_only']['ms_per_step'], d['losses'])"
  XMC_FOLD_SIGMA=0 XMC_FUSE_OPT=0 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fold0 fuse0        ', d['ms_per_step'], d['gd_only']['ms_per_step'], d['losses'])"
  XMC_PHASE_PX128=0 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('px128 off          ', d['ms_per_step'], d['gd_only']['ms_per_step'], d['losses'])"
done
$B --batch 2 --no-gd-only 2>/dev/null | tail -1 | cut -c1-300
XMC_FOLD_SIGMA=0 XMC_FUSE_OPT=0 $B --batch 2 --no-gd-only 2>/dev/null | tail -1 | cut -c1-300
PYTHONPATH=$R timeout 300 python tools/bench_phase.py --only-phase 2>&1 | grep -v amdgpu > $O/phase_px128_on.txt
XMC_PHASE_PX128=0 PYTHONPATH=$R timeout 300 python tools/bench_phase.py --only-phase 2>&1 | grep -v amdgpu > $O/phase_px128_off.txt
paste <(awk '{print $1,$2,$3,$9,$17}' $O/phase_px128_on.txt) <(awk '{print $9,$17}' $O/phase_px128_off.txt) | column -t
timeout 900 bash tools/phase_abl.sh > $O/phase_abl.txt 2>&1 < /dev/null
cat $O/phase_abl.txt
for a in "--config c3 --batch 32" "--config c3 --batch 32 --fp8" "--config c3 --batch 32 --fp8 --serial" "--config c1 --batch 56 --fp8"; do
  echo "== poison_check $a"; PYTHONPATH=$R timeout 400 python tools/poison_check.py $a 2>&1 | grep -v amdgpu | tail -14
done
