#!/bin/bash
# fused sigma(r, z) kernel: what each stage costs (-DCP_SIGMA_ABLATE bits: 1 no P(k), 2 no spline, 4 no stores, 8 no FFT; wrong results) and the samples per
# iteration of the P(k) evaluation (-DCP_SIGMA_ILP) -- variants built BESIDE the shipped library (tools/variant_lib.sh).  bash tools/sigma_ablate.sh
for flags in "" "-DCP_SIGMA_ILP=1" "-DCP_SIGMA_ILP=4" "-DCP_SIGMA_ABLATE=1" "-DCP_SIGMA_ABLATE=2" "-DCP_SIGMA_ABLATE=4" "-DCP_SIGMA_ABLATE=8" "-DCP_SIGMA_ABLATE=5" "-DCP_SIGMA_ABLATE=7"; do
  bash tools/variant_lib.sh /tmp/cp_sigma_ablate.so "$flags" cp_sigma.hip || continue
  echo "== flags: $flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_sigma_ablate.so python tools/bench_config3_streams.py 2>&1 | grep -E "fused" | tail -1
done
