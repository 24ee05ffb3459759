#!/bin/bash
# gpurun -- bash tools/splice_ablate.sh
for m in 0 1 2 4 7; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_SPLICE_ABLATE=$m -o /tmp/smb$m tools/splice_microbench.hip 2>&1 | grep -i " error" &
done
wait
for m in 0 1 2 4 7; do timeout 60 /tmp/smb$m | tail -2; done
