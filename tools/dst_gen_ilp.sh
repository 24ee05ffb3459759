#!/bin/bash
# wallish2018 with the spectra evaluated inside the forward transform (dst_generate_kernel): samples per loop iteration of the evaluation
# (-DCP_DST_GEN_ILP), diagnostic rebuilds on the GPU box.  bash tools/dst_gen_ilp.sh
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
for ilp in ${VARIANTS:-1 2 4}; do
  ( cd cosmoprimo_amd/csrc && hipcc $base -DCP_DST_GEN_ILP=$ilp -c cp_dst.hip -o cp_dst.o && make > /dev/null 2>&1 ) || echo "build failed"
  echo "== CP_DST_GEN_ILP=$ilp"; for i in 1 2; do python tools/profile_secondary.py 4w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['wallish2018']['value'], d['wallish2018']['ms'])"; done
done
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_dst.hip -o cp_dst.o && make > /dev/null 2>&1 )
