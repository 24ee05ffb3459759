#!/bin/bash
# Usage (GPU box, repo root): bash scripts/profile_config4_valu.sh <tag>
# Vector and matrix instruction counts of the two BAO filters (one rocprofv3 --pmc pass per filter over tools/profile_secondary.py 4w / 4b:
# the untimed ramp, then 65 536 vectors in four chunks) -> gpurun_out/<tag>_config4_valu.json: instructions per vector, for the fp64 roofline
# of config 4 on the bench line.
tag=${1:-r3}
export TMPDIR=/tmp
R=$PWD
for f in 4w 4b; do
  rm -rf /tmp/pmc_$f
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/pmc_$f -- python3 $R/tools/profile_secondary.py $f > /tmp/pmc_$f.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {}
for f, name in (('4w', 'wallish2018'), ('4b', 'brieden2022')):
    acc = collections.Counter()
    ndisp = 0
    for path in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % f, recursive=True):
        for row in csv.DictReader(open(path)):
            acc[row['Counter_Name']] += float(row['Counter_Value'])
            ndisp += 1
    line = None
    for l in open('/tmp/pmc_%s.log' % f):
        if l.startswith('{'):
            line = json.loads(l)
    nvec = line['vectors_through_the_filter_in_this_process']
    out[name] = {'counters_whole_run': dict(acc), 'vectors': nvec, 'per_vector': {k: v / nvec for k, v in acc.items()}, 'bench': line[name]}
json.dump(out, open('gpurun_out/%s_config4_valu_raw.json' % tag, 'w'), indent=1)
final = {'what': 'vector (SQ_INSTS_VALU) and matrix (SQ_INSTS_MFMA) wave-instructions per P(k) vector of the two BAO filters, P(k) generation and sigma8 '
                 'normalisation included: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -- python3 tools/profile_secondary.py 4w | 4b '
                 '(scripts/profile_config4_valu.sh), whole-process counts divided by the vectors that went through the filter',
         'peaks': {'valu_wave_instructions_per_s': 256 * 4 * 2.4e9 / 4, 'mfma_f64_16x16x4_per_s': 256 * 4 * 2.4e9 / 64,
                   'note': '256 CUs x 4 SIMDs; a wave64 vector instruction issues in 4 cycles (fp64 FMA: 78.6 TFLOP/s), v_mfma_f64_16x16x4_f64 in 64; 2.4 GHz nominal'}}
for name, v in out.items():
    final[name] = {'per_vector': v['per_vector'], 'vectors': v['vectors']}
json.dump(final, open('gpurun_out/%s_config4_valu.json' % tag, 'w'), indent=1)
print(json.dumps({k: v['per_vector'] for k, v in out.items()}, indent=1))
PY
