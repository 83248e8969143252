# in-step A/B of the weight-gradient split-K target (XMC_WGRAD_TUNE: 0 = 1024 workgroups, 2 = 512, 1 = 768, 3 = 1536), interleaved
for r in 1 2 3; do for t in 0 14 6; do echo "tune $t r$r: $(XMC_WGRAD_TUNE=$t python tools/bench_step_short.py --no-gd-only 2>/dev/null | tail -1)"; done; done
