#!/bin/bash
# Config 5 (1.25 M distance samples, bg_kernel) with the lean ordinate of the quadrature (shipped) against E^2 term by term in the reference's units
# (-DCP_BG_LEAN_ORDINATE=0), the variant built BESIDE the shipped library.  bash tools/ab_bg_ordinate.sh
bash tools/variant_lib.sh /tmp/cp_bg_plain.so "-DCP_BG_LEAN_ORDINATE=0" cp_background.hip || exit 1
for pass in 1 2; do
  for lib in "" /tmp/cp_bg_plain.so; do
    echo "== ${lib:-shipped (lean ordinate)}"
    COSMOPRIMO_AMD_LIBRARY=$lib python - <<'PY'
import torch, json
import bench
import cosmoprimo_amd as cp
dev = torch.device('cuda:0')
om, w0, wa, zz = bench.config5_samples(1250000, 3, torch, dev)
r = bench.config5(torch, dev, om, w0, wa, zz, reps=20)
print('%.4f ms  %.3e samples/s  parity %.1e' % (r['ms'], r['value'], r['parity_spot_check']['max_rel_err']))
PY
  done
done
