#!/bin/bash
# the table-driven logarithm / exponential of the EH98 evaluation (cp_math.h) against the polynomial forms, on ONE box, builds in turn
# (gpurun -- bash tools/math_tables_ab.sh)
line() { python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
s = d["secondary"]
print("   headline %.4f ms (frac %.4f) | config 3: %.4f ms | 3B: %.3f ms | config 4: wallish2018 %.3e  brieden2022 %.3e vectors/s | config 5: %.4f ms" % (
    d["ms_per_step"], d["roofline"]["frac"], s["config3"]["ms"], s["config3b"]["ms"], s["config4"]["wallish2018"]["value"], s["config4"]["brieden2022"]["value"], s["config5"]["ms"]))
PY
}
# (variants built BESIDE the shipped library, tools/variant_lib.sh: the tree is never rebuilt in place)
for round in 1 2; do
  for v in ${VARIANTS:-"-DCP_K_POWER_TABLES=1" "-DCP_K_POWER_TABLES=0" "-DCP_MATH_TABLES_OFF=1"}; do
    bash tools/variant_lib.sh /tmp/cp_math_tables_ab.so "$v" cp_power.hip cp_sigma.hip cp_dst.hip || continue
    echo "== $v"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_math_tables_ab.so line
  done
done
