"""IEEE divisions (v_div_fixup_f64: one per `a / b` the compiler could not turn into a multiplication, ~30 instructions each) per kernel of the library's
sources, from their assembly: where `/` sits in device code.   python tools/isa_divisions.py [source.hip ...]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sources = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'cosmoprimo_amd', 'csrc', '*.hip')))
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
for src in sources:
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', '-S', '--cuda-device-only', '-o', tmp.name, src],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(tmp.name).read()
    parts = re.split(r'\n(\S+):\s*; @', text)
    for i in range(1, len(parts) - 1, 2):
        name, body = parts[i], parts[i + 1]
        body = body.split('.end_amdhsa_kernel')[0] if '.end_amdhsa_kernel' in body else body.split('.Lfunc_end')[0]
        ndiv = len(re.findall(r'v_div_fixup_f64', body))
        total = len([line for line in body.split('\n') if re.match(r'\s+[vs]_|\s+ds_|\s+global_|\s+buffer_', line)])
        if ndiv:
            if filt:
                name = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
            print('%-22s %-72s %3d divisions in %6d instructions' % (os.path.basename(src), name.replace('(anonymous namespace)::', '')[:72], ndiv, total))
