#!/bin/bash
# occupancy / stall experiments: one workgroup per CU vs two, with and without ablations
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>/dev/null || echo "build failed $*"; }
build -DCP_ABLATE=0 -DMB_WGS_PER_CU=1 -o /tmp/mb/a0w1 &
build -DCP_ABLATE=31 -DMB_WGS_PER_CU=1 -o /tmp/mb/a31w1 &
build -DCP_ABLATE=28 -DMB_WGS_PER_CU=1 -o /tmp/mb/a28w1 &
build -DCP_ABLATE=24 -DMB_WGS_PER_CU=1 -o /tmp/mb/a24w1 &
build -DCP_ABLATE=0 -DMB_WGS_PER_CU=2 -o /tmp/mb/a0w2 &
build -DCP_ABLATE=0 -DMB_WGS_PER_CU=4 -o /tmp/mb/a0w4 &
build -DCP_ABLATE=0 -DMB_NP=2048 -DMB_WGS_PER_CU=4 -o /tmp/mb/n2048w4 &
build -DCP_ABLATE=0 -DMB_NP=2048 -DMB_WGS_PER_CU=2 -o /tmp/mb/n2048w2 &
wait
for x in a0w1 a31w1 a28w1 a24w1 a0w2 a0w4; do echo -n "$x: "; /tmp/mb/$x 100000 20; done
for x in n2048w2 n2048w4; do echo -n "$x: "; /tmp/mb/$x 200000 20; done
