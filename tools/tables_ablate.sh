#!/bin/bash
# cp_tables_rows_direct (config 3B's first kernel): what its parts cost -- diagnostic rebuilds of the library on the GPU box, -DCP_TABLES_ABLATE bits:
# 1 no stores, 2 no exponential, 4 no z contraction, 8 no table loads.  bash tools/tables_ablate.sh
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
for bits in ${VARIANTS:-0 1 8 9 0}; do
  flags="-DCP_TABLES_ABLATE=$bits"
  ( cd cosmoprimo_amd/csrc && hipcc $base $flags -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== flags: $flags"; python tools/bench_config3b_kernels.py 2>&1 | grep -E "cp_tables" | tail -1
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 )
