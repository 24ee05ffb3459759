#!/bin/bash
# Registers, spills, scratch and static LDS of every kernel of one source file, from the code object metadata (hipcc -S, the Makefile's flags).
#   bash tools/kernel_resources.sh cp_spline.hip [extra flags] [| grep tables]
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form "$@" -S --cuda-device-only -o /tmp/kr_$$.s cosmoprimo_amd/csrc/$src || exit 1
python3 - /tmp/kr_$$.s <<'PY'
import re, subprocess, sys
text = open(sys.argv[1]).read()
for block in text.split('  - .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', block).group(1)
    get = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, block).group(1))
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    print('%-90s vgpr %3d agpr %3d spill %3d sgpr %3d scratch %4d lds %6d' % (name[:90], get('vgpr_count'), int(block.split()[0]), get('vgpr_spill_count'), get('sgpr_count'),
                                                                    get('private_segment_fixed_size'), get('group_segment_fixed_size')))
PY
rm -f /tmp/kr_$$.s
