#!/bin/bash
# second derivatives + box of wallish2018 (cp_wallish_dd_box): the recursions in registers (default) against the elimination in LDS (-DCP_DD_ELIMINATION=1),
# the variant built BESIDE the shipped library (tools/variant_lib.sh).  bash tools/dd_variants.sh
python -m pytest tests/test_fused_kernels_gpu.py tests/test_bao_gpu.py -x -q -k "dd_box or wallish" 2>&1 | tail -3
echo "== recursions, launch bounds (256, 2)"; python tools/bench_dd_box.py
bash tools/variant_lib.sh /tmp/cp_dd_elimination.so "-DCP_DD_ELIMINATION=1" cp_bao.hip || exit 1
echo "== elimination"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_dd_elimination.so python tools/bench_dd_box.py
