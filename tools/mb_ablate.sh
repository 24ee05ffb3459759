#!/bin/bash
# Build and run the ablation variants of tools/fftlog_microbench.hip (on the GPU box): bash tools/mb_ablate.sh "0 1 2 4 ..."
masks=${1:-"0 1 2 4 8 16 32 24 28 31 63"}
mkdir -p /tmp/mb
for m in $masks; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_ABLATE=$m -o /tmp/mb/mb$m tools/fftlog_microbench.hip 2>/dev/null || echo "build failed $m" ) &
done
wait
for m in $masks; do /tmp/mb/mb$m 100000 20; done
