#!/bin/bash
python -m pytest tests/test_linop_gpu.py tests/test_sigma_tables_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_config3b.py 2>&1 | tail -1
python tools/bench_config3b.py 2>&1 | tail -1
