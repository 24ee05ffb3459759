#!/bin/bash
# fine-grained stamps (CP_STAMPS): where each phase of the flagship kernel spends its time
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>&1 | grep -E "error" ; }
build -DCP_STAMPS -o /tmp/mb/s0 &
build -DCP_STAMPS -DMB_WGS_PER_CU=1 -o /tmp/mb/s0w1 &
build -o /tmp/mb/m0 &
wait
for x in m0 s0 s0w1; do echo "== $x"; /tmp/mb/$x 100000 5; done
