#!/bin/bash
# Register use of every kernel of the library (development aid): name, VGPRs, spilled VGPRs, scratch bytes.  bash tools/check_spills.sh
cd "$(dirname "$0")/../cosmoprimo_amd/csrc"
for f in cp_background.hip cp_dst.hip cp_interp.hip cp_power.hip cp_rows.hip cp_spline.hip cp_fftlog_large.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -S --cuda-device-only -o /tmp/_spill.s $f 2>/dev/null
  grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size)|^\s+\.name:" /tmp/_spill.s | paste - - - - | awk -v f=$f '{print f, $2, "scratch", $4, "vgpr", $6, "spills", $8}'
done
