"""Instruction census per basic block of one kernel of an assembly listing (hipcc -S --cuda-device-only): python tools/isa_blocks.py file.s name-substring [min]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
sub = sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 25
start = [m.start() for m in re.finditer(r'^(\S*%s\S*):' % re.escape(sub), text, re.M)][0]
body = text[start:text.index('.end_amdhsa_kernel', start)]
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
total = Counter()
for k in range(-1, len(blocks) - 1, 2):
    name, code = ('entry', blocks[0]) if k < 0 else (blocks[k], blocks[k + 1])
    ins = [l.strip().split()[0] for l in code.split('\n') if l.strip() and not l.strip().startswith((';', '.', '//')) and not l.strip().endswith(':')]
    c = Counter()
    for x in ins:
        if x.startswith('v_mfma'): c['mfma'] += 1
        elif x.startswith('v_') and 'f64' in x: c['valu_f64'] += 1
        elif x.startswith('v_'): c['valu_other'] += 1
        elif x.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif x.startswith('s_barrier'): c['barrier'] += 1
        elif x.startswith('s_'): c['salu'] += 1
        elif x.startswith('ds_'): c['lds'] += 1
        elif x.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
        else: c['other'] += 1
    total.update(c)
    if len(ins) >= least:
        print('%-12s %5d  %s' % (name, len(ins), dict(c)))
print('all blocks', sum(total.values()), dict(total))
