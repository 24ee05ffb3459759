#!/bin/bash
# wallish_full_kernel with the forward side's arguments read through the kernel-argument segment where they are used (shipped) against the by-value
# parameter held in scalar registers over the loop (-DCP_FULL_KERNARG_RELOAD=0): variant built BESIDE the shipped library.  bash tools/ab_full_kernarg.sh
bash tools/variant_lib.sh /tmp/cp_full_byvalue.so "-DCP_FULL_KERNARG_RELOAD=0" cp_dst.hip || exit 1
for pass in 1 2; do
  for lib in "" /tmp/cp_full_byvalue.so; do
    echo "== ${lib:-shipped (arguments re-read)}"
    COSMOPRIMO_AMD_LIBRARY=$lib python - <<'PY'
import torch, warnings
warnings.simplefilter('ignore')
import bench
import cosmoprimo_amd as cp
dev = torch.device('cuda:0')
c4 = bench.config4(cp, torch, dev, bench.eh_parameters(125000, 2, torch, dev))
print('wallish2018 %.3f ms (%.3e/s, parity %.1e) | brieden2022 %.3f ms' % (c4['wallish2018']['ms'], c4['wallish2018']['value'], c4['wallish2018']['parity_spot_check']['max_rel_err'], c4['brieden2022']['ms']))
PY
  done
done
