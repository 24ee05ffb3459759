#!/bin/bash
# generic front/back ends at P = 16 (256 threads, 2 waves/SIMD) vs P = 8 (512 threads, radix-8 passes, 4 waves/SIMD)
mkdir -p /tmp/mb
for P in 8 16; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMB_P=$P -DMB_GENERIC=1 -o /tmp/mb/g$P tools/fftlog_microbench.hip 2>&1 | grep error & done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mb/m0 tools/fftlog_microbench.hip 2>&1 | grep error &
wait
for x in m0 g16 g8; do echo "== $x"; /tmp/mb/$x 100000 20; done
