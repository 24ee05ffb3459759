#!/bin/bash
# the inverse DST kernel with parts left out (-DCP_DST_ABLATE bits: 1 no fused map, 2 no in-place loads, 4 no stores, 8 no transform)
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
for bits in ${VARIANTS:-0 1 2 4 8 15 0}; do
  ( cd cosmoprimo_amd/csrc && hipcc $base -DCP_DST_ABLATE=$bits -c cp_dst.hip -o cp_dst.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== -DCP_DST_ABLATE=$bits"; python tools/bench_dst.py 2>/dev/null | head -1
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_dst.hip -o cp_dst.o && make > /dev/null 2>&1 )
