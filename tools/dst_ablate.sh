#!/bin/bash
# the inverse DST kernel with parts left out (-DCP_DST_ABLATE bits: 1 no fused map, 2 no in-place loads, 4 no stores, 8 no transform; wrong results):
# variants built BESIDE the shipped library (tools/variant_lib.sh), which stays as it is.  bash tools/dst_ablate.sh
for bits in ${VARIANTS:-0 1 2 4 8 15 0}; do
  bash tools/variant_lib.sh /tmp/cp_dst_ablate.so "-DCP_DST_ABLATE=$bits" cp_dst.hip || continue
  echo "== -DCP_DST_ABLATE=$bits"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_dst_ablate.so python tools/bench_dst.py 2>/dev/null | head -1
done
