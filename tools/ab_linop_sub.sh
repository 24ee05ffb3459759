#!/bin/bash
# config 3B and the operator benchmarks with and without the per-16-query windows of linop_mfma_kernel<true>, on ONE box (boxes differ by 3 %):
# the library is rebuilt with -DCP_LINOP_SUB_FRACTION=0 (never) and 0.67 (the default).   gpurun -- bash tools/ab_linop_sub.sh
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {
    rm -f cosmoprimo_amd/csrc/cp_spline.o
    make -C cosmoprimo_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $1" > /dev/null 2>&1 || { echo "build failed: $1"; return; }
    echo "== $1"
    for i in 1 2 3; do timeout 300 python tools/bench_config3b.py 2>&1 | grep "config 3B"; done
    timeout 300 python tools/bench_linop.py 2>&1 | grep "radii\|k contraction\|1024 x 1024"
}
{
    run "-DCP_LINOP_SUB_FRACTION=0."
    run "-DCP_LINOP_SUB_FRACTION=0.67"
    run "-DCP_LINOP_SUB_FRACTION=0."
    run "-DCP_LINOP_SUB_FRACTION=0.67"
} 2>&1 | tee gpurun_out/ab_linop_sub.txt
