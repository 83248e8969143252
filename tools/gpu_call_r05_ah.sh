# last check of the round: the whole GPU suite + smoke on the final tree
R=$PWD; O=$R/gpurun_out/ah; mkdir -p $O
export PYTHONPATH=$R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
