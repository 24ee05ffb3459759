#!/bin/bash
# cp_tables_rows_direct: waves per SIMD x table entries fetched a step ahead -- variants built BESIDE the shipped library (tools/variant_lib.sh).
# gpurun -- bash tools/tables_variants.sh
python -m pytest tests/test_sigma_tables_gpu.py tests/test_interpolator_contracts_gpu.py -x -q 2>&1 | tail -2
for f in "-DCP_TABLES_WAVES=3 -DCP_TABLES_PREFETCH=0" "-DCP_TABLES_WAVES=2 -DCP_TABLES_PREFETCH=1" "-DCP_TABLES_WAVES=2 -DCP_TABLES_PREFETCH=0" "-DCP_TABLES_WAVES=4 -DCP_TABLES_PREFETCH=0"; do
  bash tools/variant_lib.sh /tmp/cp_tables_variant.so "$f" cp_spline.hip || continue
  echo "== $f"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_tables_variant.so python tools/bench_config3b_kernels.py 2>&1 | grep tables
done
