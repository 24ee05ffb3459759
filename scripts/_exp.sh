#!/bin/bash
python -m pytest tests/test_background_gpu.py tests/test_ncdm_gpu.py tests/test_full_size_gpu.py tests/test_cosmology_gpu.py tests/test_fiducial_gpu.py tests/test_calculator_gpu.py -m gpu -x -q 2>&1 | tail -3
python tools/profile_secondary.py 5 | grep -o '"value": [0-9.]*\|"frac": [0-9.]*'
python tools/profile_secondary.py 5 | grep -o '"value": [0-9.]*\|"frac": [0-9.]*'
