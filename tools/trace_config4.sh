#!/bin/bash
# config 4: the kernels of ONE 16 384-vector chunk of each filter, in launch order, with their durations (rocprofv3 kernel trace of bench.config4)
export TMPDIR=/tmp
for engine in wallish2018 brieden2022; do
cat > /tmp/c4t.py <<PY
import sys
sys.path.insert(0, '.')
import torch, bench
import cosmoprimo_amd as cp
dev = torch.device('cuda', 0)
bench.RAMP_S = 0.02
par = bench.eh_parameters(2 * 16384, 2, torch, dev)
import cosmoprimo_amd.bao_filter as bf
keep = {'$engine'}
orig = bench.config4
# only the chosen engine
import warnings
from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
fid = cp.Cosmology(engine='eisenstein_hu')
state = {}
def run(sl):
    cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sl] for name, v in par.items()})
    interp = cosmo.get_fourier().pk_interpolator(z=__import__('numpy').array([0.]))
    kw = dict(cosmo_fid=fid, cosmo=cosmo) if '$engine' == 'brieden2022' else {}
    if 'filter' not in state:
        state['filter'] = PowerSpectrumBAOFilter(interp, engine='$engine', **kw)
    else:
        state['filter'](interp, cosmo=cosmo if kw else None)
    return state['filter']._pknow_rows
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    for i in range(3):
        run(slice(0, 16384)); torch.cuda.synchronize()
    torch.cuda.synchronize()
    z = torch.zeros(1, device=dev); z.add_(1.)      # marker
    run(slice(16384, 32768)); torch.cuda.synchronize()
PY
rm -rf /tmp/c4trace; rocprofv3 --kernel-trace --output-format csv -d /tmp/c4trace -- python3 /tmp/c4t.py > /tmp/c4t.log 2>&1
f=$(find /tmp/c4trace -name "*kernel_trace.csv" | head -1)
echo "==== $engine"
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last chunk: everything after the last 'FillFunctor'/'add' marker pair -- take the kernels after the last big gap (> 200 us)
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']) > 150000:
        cut = i
last = rows[cut:]
t0 = int(last[0]['Start_Timestamp'])
tot = native = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    tot += d
    if 'at::native' in name or 'rocclr' in name:
        native += d
    print('%8.1f us +%7.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, d, name[:120]))
print('kernels %d, kernel time %.1f us, of which torch glue %.1f us (%.1f %%); span %.1f us' % (len(last), tot, native, 100 * native / tot, (int(last[-1]['End_Timestamp']) - t0) / 1e3))
PY
done
