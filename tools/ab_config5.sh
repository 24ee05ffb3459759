#!/bin/bash
# Same-box A/B of two builds of the library on configs 3 and 5 and the sigma8 normalisation (as bench.py times them), builds in turn.
#   bash tools/ab_config5.sh <before.so> [rounds]
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
r3 = bench.config3(cp, torch, dev, reps=20)
om, w0, wa, zz = bench.config5_samples(1250000, 3, torch, dev)
r5 = bench.config5(torch, dev, om, w0, wa, zz, reps=30)
print('%-8s config 3 %.4f ms | config 5 %.4f ms (%.3e/s)' % (sys.argv[1], r3['ms'], r5['ms'], r5['value']))
PY
  done
done
