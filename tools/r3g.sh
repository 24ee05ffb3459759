#!/bin/bash
# fused sigma kernel: samples per iteration of the P(k) evaluation (library rebuilt on the box for each)
for ilp in 2 4 1; do
  (cd cosmoprimo_amd/csrc && touch cp_sigma.hip && make -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DCP_SIGMA_ILP=$ilp" > /dev/null 2>&1)
  echo "ILP $ilp"; python tools/bench_config3_streams.py 2>&1 | grep -E "fused|separate"
done
