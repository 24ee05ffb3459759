#!/bin/bash
# headline kernel (tools/fftlog_microbench.hip): what the 32 registers of the pre / post factors would buy as pinned pass-0 twiddles -- an upper bound, the
# factors made compile-time constants (wrong results).   bash tools/mb_pin_factors.sh
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>&1 | grep -E "error|spill" ; }
build -o /tmp/mb/p7 &
build -DCP_DIAG_CONST_FACTORS -o /tmp/mb/c7 &
build -DCP_DIAG_CONST_FACTORS -DCP_PIN_TW0=10 -o /tmp/mb/c10 &
build -DCP_DIAG_CONST_FACTORS -DCP_PIN_TW0=12 -o /tmp/mb/c12 &
build -DCP_DIAG_CONST_FACTORS -DCP_PIN_TW0=14 -o /tmp/mb/c14 &
build -DCP_DIAG_CONST_FACTORS -DCP_PIN_TW0=15 -o /tmp/mb/c15 &
wait
for rnd in 1 2; do for x in p7 c7 c10 c12 c14 c15; do echo "== $x"; /tmp/mb/$x 100000 20 | tail -1; done; done
