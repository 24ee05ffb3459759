cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -25
