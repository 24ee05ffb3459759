cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
( bash tools/mb_variants.sh "default:-DCP_X=0" "noscreen:-DCP_ROW_SCREEN=0" ) > gpurun_out/r2/exp11_mb.log 2>&1
cat gpurun_out/r2/exp11_mb.log
timeout 900 python -m pytest tests/test_fftlog_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
