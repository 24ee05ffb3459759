#!/bin/bash
# fused sigma(r, z) kernel, round 6: what is left when stages are taken out (-DCP_SIGMA_ABLATE bits: 1 no P(k), 2 no spline, 4 no stores, 8 no FFT; wrong
# results), builds BESIDE the shipped library (tools/variant_lib.sh).   bash tools/sigma_ablate_spline.sh
for flags in ${VARIANTS:-0 2 11 4 0 11}; do
  bash tools/variant_lib.sh /tmp/cp_sigma_ablate.so "-DCP_SIGMA_ABLATE=$flags" cp_sigma.hip || continue
  echo "== -DCP_SIGMA_ABLATE=$flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_sigma_ablate.so python tools/time_config3_raw.py 2>&1 | grep "config 3 call" | tail -2
done
