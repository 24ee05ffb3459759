"""Where a kernel talks to memory and where it waits for it, in program order, mapped to source lines -- the reading of the ISA behind round 6's findings
(profiles/r6_geospline_prefilter.txt): flat / global / buffer loads (L), stores (S), workgroup barriers (B) and every s_waitcnt on the vector-memory counter
(W<n>: at most n operations may still be in flight; on gfx950 loads AND stores retire through this one counter, in order).  What to look for:
  * `L W0 L W0 ...`: a load-use-load chain -- each load a memory round trip of its own (plan entries requested block by block behind wave-uniform branches);
  * `S ... L ... W0` inside a loop: a load issued behind stores waits for their acknowledgement;
  * flat_load where a scalar load was meant (a constant read through a laundered generic pointer): see load_uniform (csrc/cp_power_eval.h).
No GPU needed (hipcc -S -gline-tables-only with the Makefile's flags).
    python tools/isa_waits.py cp_sigma.hip geospline_kernelILb1ELi4E          # kernels whose mangled name contains the pattern
    python tools/isa_waits.py cp_dst.hip wallish_full_kernelILi49ELi0 -DCP_TAIL_ROT_BATCH=1"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'cosmoprimo_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', '-gline-tables-only', '-S', '--cuda-device-only']


def trace(text, pattern):
    files = dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', text))
    files.update(dict((int(a), b) for a, b in re.findall(r'\.file\s+(\d+)\s+"([^"]+)"\s*$', text, re.M)))
    for m in re.finditer(r'^(_Z\w+):\s*; @', text, re.M):
        name = m.group(1)
        if pattern not in name:
            continue
        body = text[m.end():text.index('s_endpgm', m.end())]
        where, events = '?', []
        for line in body.splitlines():
            s = line.strip()
            loc = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
            if loc:
                where = '%s:%s' % (os.path.basename(files.get(int(loc.group(1)), loc.group(1))), loc.group(2))
            elif re.match(r'(global|flat|buffer)_load', s):
                events.append((where, 'FLAT-L' if s.startswith('flat') else 'L'))
            elif re.match(r'(global|flat|buffer)_store', s):
                events.append((where, 'S'))
            elif s.startswith('s_barrier'):
                events.append(('', 'B'))
            elif s.startswith('s_waitcnt') and 'vmcnt' in s:
                events.append((where, 'W' + re.search(r'vmcnt\((\d+)\)', s).group(1)))
        out, prev, count = [], None, 0
        for e in events + [None]:
            if e == prev:
                count += 1
                continue
            if prev is not None:
                out.append('%s%s%s' % (prev[1], 'x%d' % count if count > 1 else '', ' @' + prev[0] if prev[0] else ''))
            prev, count = e, 1
        print('== %s' % name)
        print(' | '.join(out))


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    source, pattern, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        subprocess.run(['hipcc'] + FLAGS + extra + ['-o', tmp.name, source], check=True, capture_output=True, cwd=CSRC)
        trace(open(tmp.name).read(), pattern)


if __name__ == '__main__':
    main()
