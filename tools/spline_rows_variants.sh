#!/bin/bash
# spline_rows_kernel: what its parts cost (CP_ROWS_ABLATE, cp_spline_rows.hip), the library rebuilt for each variant on the GPU box
#   gpurun -- bash tools/spline_rows_variants.sh        -> gpurun_out/spline_rows_variants.txt (profiles/r3_spline_rows_ablation.txt)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {
    rm -f cosmoprimo_amd/csrc/cp_spline_rows.o
    make -C cosmoprimo_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $1" > /dev/null 2>&1 || { echo "build failed: $1"; return; }
    echo "== $1"
    timeout 300 python tools/bench_spline_rows.py 2>&1 | grep "10000 x 64"
}
{
    for a in 0 1 2 3 4 7 8 16 32 24 56; do run "-DCP_ROWS_ABLATE=$a"; done
} 2>&1 | tee gpurun_out/spline_rows_variants.txt
