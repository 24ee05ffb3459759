#!/bin/bash
# shader clock (s_memtime ticks per ms) of the ablation builds: is the kernel power / current limited?
mkdir -p /tmp/mb
for m in 0 1 4 8 16 24 28 31 32 33 63; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_STAMPS -DCP_ABLATE=$m -o /tmp/mb/c$m tools/fftlog_microbench.hip 2>&1 | grep error &
done
wait
for m in 0 1 4 8 16 24 28 31 32 33 63; do /tmp/mb/c$m 100000 10 | grep -E "ablate|total=" | sed -E 's/per pair per wave.*total=/   total=/' ; done
