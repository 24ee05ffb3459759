cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
( bash tools/mb_variants.sh "xor:-DCP_PADDED_LDS=0" "padperm:-DCP_PADDED_LDS=1" "xor_wide:-DCP_PADDED_LDS=0 -DCP_WIDE_IO=1" "padpermstamps:-DCP_STAMPS" ) > gpurun_out/r2/exp4_mb.log 2>&1
cat gpurun_out/r2/exp4_mb.log
timeout 900 python -m pytest tests/test_fftlog_gpu.py tests/test_full_size_gpu.py tests/test_dst_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r2/exp4_tests.log
