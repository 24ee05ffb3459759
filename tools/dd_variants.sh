base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
python -m pytest tests/test_fused_kernels_gpu.py tests/test_bao_gpu.py -x -q -k "dd_box or wallish" 2>&1 | tail -3
echo "== recursions, launch bounds (256, 2)"; python tools/bench_dd_box.py
( cd cosmoprimo_amd/csrc && hipcc $base -DCP_DD_ELIMINATION=1 -c cp_bao.hip -o cp_bao.o && make > /dev/null 2>&1 )
echo "== elimination"; python tools/bench_dd_box.py
