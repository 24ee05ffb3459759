"""Instruction census of the fused FFTLog kernels from a hipcc -S --cuda-device-only dump (development aid).
   python tools/isa_census.py /tmp/g3.s [name-substring]"""
import collections
import re
import sys

src = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ''
parts = re.split(r'\n(_ZN5cpfft\w+): +; @\w+\n', src)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if want not in name:
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/' or line.endswith(':'):
            continue
        c[line.split()[0]] += 1
    grp = lambda f: sum(v for k, v in c.items() if f(k))  # noqa: E731
    print(name)
    print('  total %d | valu %d (f64 %d, dpp %d) | salu %d | ds %d (read %d, write %d) | buffer %d | s_barrier %d | s_waitcnt %d | s_nop %d | scratch %d | lane spills %d' % (
        sum(c.values()), grp(lambda k: k.startswith('v_')), grp(lambda k: 'f64' in k), grp(lambda k: 'dpp' in k), grp(lambda k: k.startswith('s_')),
        grp(lambda k: k.startswith('ds_')), grp(lambda k: k.startswith('ds_read')), grp(lambda k: k.startswith('ds_write')), grp(lambda k: k.startswith('buffer')),
        c['s_barrier'], c['s_waitcnt'], c['s_nop'], grp(lambda k: 'scratch' in k), c['v_readlane_b32'] + c['v_writelane_b32']))
    if want:
        print('  ', dict(c.most_common(45)))


def loop_census(path, want):
    """census of the largest loop (label .. backward branch) of the first kernel whose name contains `want`"""
    src = open(path).read()
    parts = re.split(r'\n(_ZN5cpfft\w+): +; @\w+\n', src)
    for i in range(1, len(parts), 2):
        if want not in parts[i]:
            continue
        lines = parts[i + 1].split('.Lfunc_end')[0].split('\n')
        labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        best = (0, 0)
        for n, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and labels.get(m.group(1), n) < n and n - labels[m.group(1)] > best[1] - best[0]:
                best = (labels[m.group(1)], n)
        c = collections.Counter()
        for l in lines[best[0]:best[1] + 1]:
            l = l.strip()
            if l and l[0] not in '.;/' and not l.endswith(':'):
                c[l.split()[0]] += 1
        return c


if __name__ == '__main__' and len(sys.argv) > 2:
    c = loop_census(sys.argv[1], sys.argv[2])
    print('main loop: total %d, valu %d (f64 %d), salu %d, ds %d, vmem %d' % (
        sum(c.values()), sum(v for k, v in c.items() if k.startswith('v_')), sum(v for k, v in c.items() if 'f64' in k),
        sum(v for k, v in c.items() if k.startswith('s_')), sum(v for k, v in c.items() if k.startswith('ds_')),
        sum(v for k, v in c.items() if k.startswith('buffer'))))
    print('  salu', {k: v for k, v in c.most_common() if k.startswith('s_')})
    print('  valu', {k: v for k, v in c.most_common() if k.startswith('v_') and 'f64' not in k})
