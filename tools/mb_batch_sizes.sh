#!/bin/bash
# batch-size sweep of the flagship kernel (launch overhead / tail effects) and the no-HBM ablation at two sizes
mkdir -p /tmp/mb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" tools/fftlog_microbench.hip 2>/dev/null || echo "build failed $*"; }
build -o /tmp/mb/p0 &
build -DCP_ABLATE=24 -o /tmp/mb/a24 &
wait
for n in 2048 4096 8192 16384 32768 100000 400000; do /tmp/mb/p0 $n 50; done
for n in 8192 100000; do /tmp/mb/a24 $n 50; done
rocm-smi --showclocks 2>/dev/null | head -20
