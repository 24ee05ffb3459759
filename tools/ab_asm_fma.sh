#!/bin/bash
# The one-instruction Horner step (asm v_fma_f64 in the polynomials of cp_math.h: shipped) against the compiler's v_fmac + copy (-DCP_ASM_FMA=0), on the kernels that
# evaluate EH98 / the distance quadrature: variant built BESIDE the shipped library.  bash tools/ab_asm_fma.sh
bash tools/variant_lib.sh /tmp/cp_asmfma.so "-DCP_ASM_FMA=${ASM:-0}" cp_background.hip cp_sigma.hip cp_power.hip cp_dst.hip cp_pipeline.hip || exit 1
for pass in 1 2; do
  for lib in "" /tmp/cp_asmfma.so; do
    echo "== ${lib:-shipped}"
    COSMOPRIMO_AMD_LIBRARY=$lib python - <<'PY'
import torch, warnings
warnings.simplefilter('ignore')
import bench
import cosmoprimo_amd as cp
dev = torch.device('cuda:0')
r3 = bench.config3(cp, torch, dev, reps=20)
om, w0, wa, zz = bench.config5_samples(1250000, 3, torch, dev)
r5 = bench.config5(torch, dev, om, w0, wa, zz, reps=20)
c4 = bench.config4(cp, torch, dev, bench.eh_parameters(125000, 2, torch, dev))
print('config 3 %.4f ms | config 5 %.4f ms | wallish2018 %.3f ms (%.3e/s) | brieden2022 %.3f ms (%.3e/s)' % (r3['ms'], r5['ms'], c4['wallish2018']['ms'], c4['wallish2018']['value'],
      c4['brieden2022']['ms'], c4['brieden2022']['value']))
PY
  done
done
