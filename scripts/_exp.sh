cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_linop_gpu.py tests/test_xi_gpu.py tests/test_bao2_gpu.py tests/test_bao_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_linop -- python3 tools/bench_linop.py > gpurun_out/r2/prof_linop.log 2>&1
grep -v amdgpu gpurun_out/r2/prof_linop.log | head -6
