#!/bin/bash
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>/dev/null || echo "build failed $*"; }
for e in 1 2 3 4 5; do build -DCP_EXP_PRIO=$e -o /tmp/mb/p$e & done
build -o /tmp/mb/p0 &
wait
for r in 1 2; do for e in 0 1 2 3 4 5; do echo -n "prio$e: "; /tmp/mb/p$e 100000 20; done; done
