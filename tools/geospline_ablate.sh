#!/bin/bash
# FFTLog + solved spline (fftlog_geospline_kernel): what the parts of its tail cost -- variants built BESIDE the shipped library (tools/variant_lib.sh; wrong
# results), -DCP_GEO_ABLATE bits: 1 no carries between lanes, 2 one query per lane, 4 no root, 8 no solve at all, 16 no stores, 32 queries made up in
# registers.  bash tools/geospline_ablate.sh
for bits in ${VARIANTS:-0 1 2 4 16 32 7 8}; do
  flags="-DCP_GEO_ABLATE=$bits"
  bash tools/variant_lib.sh /tmp/cp_geo_ablate.so "$flags" cp_sigma.hip || continue
  echo "== flags: $flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_geo_ablate.so python tools/bench_geospline.py 2>&1 | grep -E "geospline|operator" | tail -4
done
