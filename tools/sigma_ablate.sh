#!/bin/bash
# fused sigma(r, z) kernel: what each stage costs (diagnostic rebuilds of the library on the GPU box; -DCP_SIGMA_ABLATE bits: 1 no P(k), 2 no spline,
# 4 no stores, 8 no FFT) and the samples per iteration of the P(k) evaluation (-DCP_SIGMA_ILP)
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
for flags in "" "-DCP_SIGMA_ILP=1" "-DCP_SIGMA_ILP=4" "-DCP_SIGMA_ABLATE=1" "-DCP_SIGMA_ABLATE=2" "-DCP_SIGMA_ABLATE=4" "-DCP_SIGMA_ABLATE=8" "-DCP_SIGMA_ABLATE=5" "-DCP_SIGMA_ABLATE=7"; do
  ( cd cosmoprimo_amd/csrc && hipcc $base $flags -c cp_sigma.hip -o cp_sigma.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== flags: $flags"; python tools/bench_config3_streams.py 2>&1 | grep -E "fused" | tail -1
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_sigma.hip -o cp_sigma.o && make > /dev/null 2>&1 )
