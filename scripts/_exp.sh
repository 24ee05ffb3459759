cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/r2/exp8_tests.log
tail -5 gpurun_out/r2/exp8_tests.log
