#!/bin/bash
# fused sigma(r, z) kernel: the second row of a pair's results stored behind the NEXT pair's wait for U (-DCP_SIGMA_DEFER=1), with and without the loads that
# sit behind that wait (diagnostic, wrong results: -DCP_SIGMA_ABLATE=32 growth factors and radii made up, -DCP_DIAG_SKIP_TW0_RELOAD no reload of the pass-0
# twiddles) -- variants built BESIDE the shipped library (tools/variant_lib.sh), timed without the parity check (tools/time_config3_raw.py).
for flags in "" "-DCP_SIGMA_DEFER=1" "-DCP_SIGMA_DEFER=1 -DCP_SIGMA_ABLATE=32 -DCP_DIAG_SKIP_TW0_RELOAD" "-DCP_SIGMA_ABLATE=32 -DCP_DIAG_SKIP_TW0_RELOAD" ""; do
  bash tools/variant_lib.sh /tmp/cp_sigma_defer.so "$flags" cp_sigma.hip > /dev/null 2>&1 || { echo "== $flags: build failed"; continue; }
  echo "== flags: $flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_sigma_defer.so python tools/time_config3_raw.py 2>&1 | grep "config 3 call" | tail -2
done
