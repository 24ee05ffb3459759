cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_linop -- python3 tools/bench_linop.py > gpurun_out/r2/prof_linop.log 2>&1
grep -v "amdgpu\|rocprofv3\|^W2026\|^E2026" gpurun_out/r2/prof_linop.log
