#!/bin/bash
# cp_tables_rows_direct (config 3B's first kernel): what its parts cost -- variants built BESIDE the shipped library (tools/variant_lib.sh; wrong results),
# -DCP_TABLES_ABLATE bits: 1 no stores, 2 no exponential, 4 no z contraction, 8 no table loads.  bash tools/tables_ablate.sh
for bits in ${VARIANTS:-0 1 8 9 0}; do
  flags="-DCP_TABLES_ABLATE=$bits"
  bash tools/variant_lib.sh /tmp/cp_tables_ablate.so "$flags" cp_spline.hip || continue
  echo "== flags: $flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_tables_ablate.so python tools/bench_config3b_kernels.py 2>&1 | grep -E "cp_tables" | tail -1
done
