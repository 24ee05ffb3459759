#!/bin/bash
# FFTLog + solved spline (fftlog_geospline_kernel): what the parts of its tail cost -- diagnostic rebuilds of the library on the GPU box,
# -DCP_GEO_ABLATE bits: 1 no carries between lanes, 2 one query per lane, 4 no root, 8 no solve at all, 16 no stores, 32 queries made up in registers.  bash tools/geospline_ablate.sh
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
for bits in ${VARIANTS:-0 1 2 4 16 32 7 8}; do
  flags="-DCP_GEO_ABLATE=$bits"
  ( cd cosmoprimo_amd/csrc && hipcc $base $flags -c cp_sigma.hip -o cp_sigma.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== flags: $flags"; python tools/bench_geospline.py 2>&1 | grep -E "geospline|operator" | tail -4
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_sigma.hip -o cp_sigma.o && make > /dev/null 2>&1 )
