#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/profile_secondary.py 4 | cut -c1-120
python tools/profile_secondary.py 4 | grep -o '"value": [0-9.]*'
python tools/profile_secondary.py 3 | grep -o '"value": [0-9.]*'
