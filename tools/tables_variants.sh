#!/bin/bash
# cp_tables_rows_direct: waves per SIMD x table entries fetched a step ahead (gpurun -- bash tools/tables_variants.sh)
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
python -m pytest tests/test_sigma_tables_gpu.py tests/test_interpolator_contracts_gpu.py -x -q 2>&1 | tail -2
for f in "-DCP_TABLES_WAVES=3 -DCP_TABLES_PREFETCH=0" "-DCP_TABLES_WAVES=2 -DCP_TABLES_PREFETCH=1" "-DCP_TABLES_WAVES=2 -DCP_TABLES_PREFETCH=0" "-DCP_TABLES_WAVES=4 -DCP_TABLES_PREFETCH=0"; do
  ( cd cosmoprimo_amd/csrc && hipcc $base $f -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== $f"; python tools/bench_config3b_kernels.py 2>&1 | grep tables
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 )
