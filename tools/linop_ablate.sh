#!/bin/bash
# gpurun -- bash tools/linop_ablate.sh
for m in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DCP_LINOP_ABLATE=$m -o /tmp/lmb$m tools/linop_microbench.hip 2>&1 | grep -i " error" &
done
wait
for m in 0 1 2 3; do timeout 60 /tmp/lmb$m | tail -1; done
