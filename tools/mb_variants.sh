#!/bin/bash
# Build and time named variants of the flagship kernel (tools/fftlog_microbench.hip) on the GPU box:
#   bash tools/mb_variants.sh "name:-DFLAG1 -DFLAG2" "name2:..."      (each built in parallel, run twice in turn)
mkdir -p /tmp/mb
names=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  names+=($name)
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -o /tmp/mb/$name tools/fftlog_microbench.hip 2>/tmp/mb/$name.err || { echo "build failed: $name"; tail -5 /tmp/mb/$name.err; } ) &
done
wait
for rep in 1 2; do
  for n in "${names[@]}"; do [ -x /tmp/mb/$n ] && { echo -n "$n: "; /tmp/mb/$n 100000 20 | head -3; }; done
done
