#!/bin/bash
# Same-box A/B of two builds of the library through the driver's command (bench.py: headline + configs 3 / 3B / 4 / 5), builds in turn.
#   bash tools/ab_round6.sh <before.so> <out-prefix> [rounds]
before=$1; out=$2; rounds=${3:-2}
mkdir -p "$(dirname "$out")"
for r in $(seq 1 $rounds); do
  COSMOPRIMO_AMD_LIBRARY=$before python bench.py --no-cpu-baseline > "${out}_before_$r.json" 2> "${out}_before_$r.err"
  python bench.py --no-cpu-baseline > "${out}_after_$r.json" 2> "${out}_after_$r.err"
done
python - "$out" "$rounds" <<'PY'
import json, sys
out, rounds = sys.argv[1], int(sys.argv[2])
def pick(d):
    sec = d.get('secondary', {})
    row = {'headline_ms': d.get('ms_per_step'), 'frac': d.get('roofline', {}).get('frac')}
    for k, v in sec.items():
        if isinstance(v, dict):
            for kk in ('ms', 'value', 'vectors_per_s', 'samples_per_s'):
                if kk in v: row['%s.%s' % (k, kk)] = v[kk]
            for kk, vv in v.items():
                if isinstance(vv, dict):
                    for k3 in ('ms', 'value'):
                        if k3 in vv: row['%s.%s.%s' % (k, kk, k3)] = vv[k3]
    return row
for which in ('before', 'after'):
    for r in range(1, rounds + 1):
        try:
            d = json.loads(open('%s_%s_%d.json' % (out, which, r)).read().strip().splitlines()[-1])
            print(which, r, json.dumps(pick(d)))
        except Exception as e:
            print(which, r, 'failed', e)
PY
