#!/bin/bash
# Same-box A/B of compile-time variants of the kernels that evaluate EH98 (cp_sigma / cp_power / cp_dst / cp_bao / cp_pipeline), each built BESIDE the
# shipped library (tools/variant_lib.sh) and run through bench.py's configs 3 / 4 / 5, two passes, builds in turn.
#   bash tools/ab_variants.sh "<name>=<flags>" ["<name>=<flags>" ...]        e.g.  "tables=-DCP_SIGMA_RZ_TABLES=1" "merged=-DCP_EH_MERGED_RECIP=1"
libs=("shipped=")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  if bash tools/variant_lib.sh /tmp/cp_var_$name.so "$flags" cp_sigma.hip cp_power.hip cp_dst.hip cp_bao.hip cp_pipeline.hip cp_background.hip; then libs+=("$name=/tmp/cp_var_$name.so"); else echo "== $name: build failed"; fi
done
for pass in 1 2; do
  for entry in "${libs[@]}"; do
    name=${entry%%=*}; lib=${entry#*=}
    COSMOPRIMO_AMD_LIBRARY=$lib python - "$name" <<'PY'
import sys, torch, warnings
warnings.simplefilter('ignore')
import bench
import cosmoprimo_amd as cp
dev = torch.device('cuda:0')
r3 = bench.config3(cp, torch, dev, reps=20)
om, w0, wa, zz = bench.config5_samples(1250000, 3, torch, dev)
r5 = bench.config5(torch, dev, om, w0, wa, zz, reps=20)
c4 = bench.config4(cp, torch, dev, bench.eh_parameters(125000, 2, torch, dev))
print('%-12s config 3 %.4f ms | config 5 %.4f ms | wallish2018 %.3f ms (%.3e/s) | brieden2022 %.3f ms (%.3e/s)' % (sys.argv[1], r3['ms'], r5['ms'], c4['wallish2018']['ms'],
      c4['wallish2018']['value'], c4['brieden2022']['ms'], c4['brieden2022']['value']))
PY
  done
done
