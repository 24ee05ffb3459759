#!/bin/bash
# sigma_rz_kernel (config 3): the per-thread constants of the P(k) evaluation kept in vector registers over the loop over pairs (spilled: reloaded from
# scratch at the top of every pair) against LDS + scalar registers; variants built beside the shipped library, alternately on one box.
bash tools/variant_lib.sh /tmp/cp_sigma_v0.so "-DCP_SIGMA_FIRST_K_IN_LDS=0" cp_sigma.hip || exit 1
bash tools/variant_lib.sh /tmp/cp_sigma_v1.so "-DCP_SIGMA_FIRST_K_IN_LDS=1" cp_sigma.hip || exit 1
for round in 1 2 3; do
  for v in 0 1; do
    echo "CP_SIGMA_FIRST_K_IN_LDS=$v: $(COSMOPRIMO_AMD_LIBRARY=/tmp/cp_sigma_v$v.so python tools/profile_secondary.py 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms'])")"
  done
done
