"""Registers, spills, scratch and static LDS of every kernel of the library from the code-object metadata (hipcc -S with the Makefile's flags), as JSON:
    python tools/kernel_resources_json.py > profiles/r6_kernel_resources.json
No GPU needed.  The library's objects are built from the same sources with the same flags (cosmoprimo_amd/csrc/Makefile)."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cosmoprimo_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', '-S', '--cuda-device-only']


def demangle(names):
    if not names:
        return names
    try:
        out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
    except OSError:
        return names
    return out if len(out) == len(names) else names


def kernels_of(source, extra=()):
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['hipcc'] + FLAGS + list(extra) + ['-o', tmp.name, source], check=True, capture_output=True, cwd=CSRC)
        text = open(tmp.name).read()
    found = []
    for block in text.split('  - .agpr_count:')[1:]:
        def get(key):
            return int(re.search(r'\.%s:\s+(\d+)' % key, block).group(1))
        found.append({'name': re.search(r'\.name:\s+(\S+)', block).group(1), 'vgpr_count': get('vgpr_count'), 'agpr_count': int(block.split()[0]),
                      'vgpr_spill_count': get('vgpr_spill_count'), 'sgpr_count': get('sgpr_count'), 'sgpr_spill_count': get('sgpr_spill_count'),
                      'private_segment_fixed_size': get('private_segment_fixed_size'), 'group_segment_fixed_size': get('group_segment_fixed_size')})
    for entry, name in zip(found, demangle([k['name'] for k in found])):
        entry['name'] = name.replace('(anonymous namespace)::', '')
    return found


def main():
    out = {'flags': ' '.join(FLAGS[:-2]), 'sources': {}}
    for path in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        base = os.path.basename(path)
        if base == 'cp_fftlog_inst.hip':      # one translation unit per size group (Makefile)
            for g in range(5):
                out['sources']['%s -DCP_INST_GROUP=%d' % (base, g)] = kernels_of(base, ['-DCP_INST_GROUP=%d' % g])
        else:
            out['sources'][base] = kernels_of(base)
    spilling = [(src, k['name'], k['vgpr_spill_count']) for src, ks in out['sources'].items() for k in ks if k['vgpr_spill_count']]
    out['kernels'] = sum(len(ks) for ks in out['sources'].values())
    out['kernels_with_spilled_vgprs'] = [{'source': s, 'name': n, 'vgpr_spill_count': c} for s, n, c in spilling]
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
