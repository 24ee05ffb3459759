#!/bin/bash
# wallish2018 with the spectra evaluated inside the forward transform (dst_generate_kernel): samples per loop iteration of the evaluation
# (-DCP_DST_GEN_ILP), variants built BESIDE the shipped library (tools/variant_lib.sh).  bash tools/dst_gen_ilp.sh
for ilp in ${VARIANTS:-1 2 4}; do
  bash tools/variant_lib.sh /tmp/cp_dst_gen_ilp.so "-DCP_DST_GEN_ILP=$ilp" cp_dst.hip || continue
  echo "== CP_DST_GEN_ILP=$ilp"; for i in 1 2; do COSMOPRIMO_AMD_LIBRARY=/tmp/cp_dst_gen_ilp.so python tools/profile_secondary.py 4w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['wallish2018']['value'], d['wallish2018']['ms'])"; done
done
