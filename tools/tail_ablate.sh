#!/bin/bash
# wallish_tail_kernel (cp_dst.hip): what its stages cost -- variants built BESIDE the shipped library (tools/variant_lib.sh; wrong results), -DCP_TAIL_ABLATE bits:
# 1 no second derivatives / box, 2 no transform, 4 no exponential, 8 no splice, 16 no stores of the result.  bash tools/tail_ablate.sh
for bits in ${VARIANTS:-0 1 2 4 8 16 31 0}; do
  bash tools/variant_lib.sh /tmp/cp_tail_ablate.so "-DCP_TAIL_ABLATE=$bits $EXTRA" cp_dst.hip || continue
  echo "== -DCP_TAIL_ABLATE=$bits $EXTRA"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_tail_ablate.so python tools/bench_wallish_tail.py 2>/dev/null | tail -1
done
