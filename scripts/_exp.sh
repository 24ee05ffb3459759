cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_fftlog_gpu.py tests/test_full_size_gpu.py tests/test_dst_gpu.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r2/exp6_tests.log
