#!/bin/bash
# gpurun -- bash tools/splice_ablate.sh        (both kernels of cp_splice_apply on 32 768 vectors; -DCP_SPLICE_UNIFORM_ABLATE bits: 1 no recursions,
# 2 no evaluation, 4 no window sums; -DCP_SPLICE_ABLATE: the elimination kernel's)
for m in 0 1 2 4 7; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_SPLICE_ABLATE=$m -DCP_SPLICE_UNIFORM_ABLATE=$m -o /tmp/smb$m tools/splice_microbench.hip 2>&1 | grep -i " error" &
done
wait
for m in 0 1 2 4 7; do timeout 120 /tmp/smb$m | tail -7; done
