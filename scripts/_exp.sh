#!/bin/bash
python -m pytest tests/test_dst_gpu.py tests/test_bao_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/profile_secondary.py 4 | grep -o '"value": [0-9.]*'
python tools/profile_secondary.py 4 | grep -o '"value": [0-9.]*'
