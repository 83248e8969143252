export PYTHONPATH=.
for t in 0 2 4 6 8; do echo "tune $t:"; XMC_WGRAD_TUNE=$t python tools/bench_phase.py --iters 3 2>&1 | grep -v amdgpu | cut -c1-22,128- | tail -11; done
