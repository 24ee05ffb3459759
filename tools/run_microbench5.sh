#!/bin/bash
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>&1 | grep -E "error" ; }
build -DCP_STAMPS -o /tmp/mb/s0 &
build -DCP_STAMPS -DMB_WGS_PER_CU=1 -o /tmp/mb/s0w1 &
build -DCP_STAMPS -DCP_ABLATE=24 -o /tmp/mb/s24 &
build -DCP_STAMPS -DCP_ABLATE=31 -o /tmp/mb/s31 &
build -DCP_STAMPS -DCP_ABLATE=31 -DMB_WGS_PER_CU=1 -o /tmp/mb/s31w1 &
wait
for x in s0 s0w1 s24 s31 s31w1; do echo "== $x"; /tmp/mb/$x 100000 5; done
