#!/bin/bash
# flagship variant (HALF_ZERO / HALF) at P = 16 (256 threads, 2 waves/SIMD) vs P = 8 (512 threads, 4 waves/SIMD)
mkdir -p /tmp/mb
for P in 8 16; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMB_P=$P -o /tmp/mb/h$P tools/fftlog_microbench.hip 2>&1 | grep error & done
wait
for x in h16 h8 h16 h8; do echo "== $x"; /tmp/mb/$x 100000 20; done
