#!/bin/bash
# library build flags vs plain -O3 on the same box, then the bench (same box): is a slower bench the box or the build?
bash tools/mb_variants.sh "plain:" "libflags:-fPIC -mllvm -amdgpu-mfma-vgpr-form" "pin0:-DCP_PIN_TW0=0" "pin0nont:-DCP_PIN_TW0=0 -DMB_STREAM_ROWS=0"
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null
