#!/bin/bash
# Same-box A/B of two builds of the library on config 4 (125 000 EH98 vectors through both filters, as bench.py times them), builds in turn.
#   bash tools/ab_config4.sh <before.so> [rounds]
before=$1; rounds=${2:-3}
for r in $(seq 1 $rounds); do
  for entry in "before=$before" "after="; do
    name=${entry%%=*}; lib=${entry#*=}
    COSMOPRIMO_AMD_LIBRARY=$lib python - "$name" <<'PY'
import sys, torch, warnings
warnings.simplefilter('ignore')
import bench
import cosmoprimo_amd as cp
dev = torch.device('cuda:0')
c4 = bench.config4(cp, torch, dev, bench.eh_parameters(125000, 2, torch, dev))
print('%-8s wallish2018 %.3f ms (%.3e/s) | brieden2022 %.3f ms (%.3e/s)' % (sys.argv[1], c4['wallish2018']['ms'], c4['wallish2018']['value'], c4['brieden2022']['ms'], c4['brieden2022']['value']))
PY
  done
done
