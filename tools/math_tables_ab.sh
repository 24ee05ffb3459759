#!/bin/bash
# the table-driven logarithm / exponential of the EH98 evaluation (cp_math.h) against the polynomial forms, on ONE box, builds in turn
# (gpurun -- bash tools/math_tables_ab.sh)
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
line() { python - <<PY
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
d = json.loads(out)
s = d["secondary"]
print("   headline %.4f ms (frac %.4f) | config 3: %.4f ms | 3B: %.3f ms | config 4: wallish2018 %.3e  brieden2022 %.3e vectors/s | config 5: %.4f ms" % (
    d["ms_per_step"], d["roofline"]["frac"], s["config3"]["ms"], s["config3b"]["ms"], s["config4"]["wallish2018"]["value"], s["config4"]["brieden2022"]["value"], s["config5"]["ms"]))
PY
}
build() { ( cd cosmoprimo_amd/csrc && for f in cp_power cp_sigma cp_dst; do hipcc $base $1 -c $f.hip -o $f.o & done; wait; make > /dev/null 2>&1 ); }
for round in 1 2; do
  for v in ${VARIANTS:-"-DCP_K_POWER_TABLES=1" "-DCP_K_POWER_TABLES=0" "-DCP_MATH_TABLES_OFF=1"}; do
    build "$v"; echo "== $v"; line
  done
done
build ""
