#!/bin/bash
# Usage (GPU box, repo root): bash scripts/profile_config4_valu.sh <tag>
# Vector and matrix instruction counts of the two BAO filters (one rocprofv3 --pmc pass per filter over tools/profile_secondary.py 4w / 4b:
# one untimed chunk, then bench.CONFIG4_PROFILE_CHUNKS timed chunks of bench.CONFIG4_CHUNK vectors) -> gpurun_out/<tag>_config4_valu.json: wave-instructions per vector
# of the PACKAGE'S OWN kernels (framework and runtime kernels are listed apart), kernel by kernel, stamped with the library they were taken on.
tag=${1:-r4}
export TMPDIR=/tmp
R=$PWD
for f in 4w 4b; do
  rm -rf /tmp/pmc_$f
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/pmc_$f -- python3 $R/tools/profile_secondary.py $f > /tmp/pmc_$f.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, hashlib, json, sys, collections
tag = sys.argv[1]
lib = hashlib.sha256(open('cosmoprimo_amd/libcosmoprimo_amd.so', 'rb').read()).hexdigest()[:16]
final = {'what': 'vector (SQ_INSTS_VALU) and matrix (SQ_INSTS_MFMA) wave-instructions per P(k) vector of the two BAO filters, P(k) generation and sigma8 '
                 'normalisation included, kernels of the package only: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -- python3 '
                 'tools/profile_secondary.py 4w | 4b (scripts/profile_config4_valu.sh), per-kernel sums over the process divided by the vectors that went through the filter',
         'library_sha256_16': lib,
         'peaks': {'valu_wave_instructions_per_s': 256 * 4 * 2.4e9 / 4, 'mfma_f64_16x16x4_per_s': 256 * 4 * 2.4e9 / 64,
                   'note': '256 CUs x 4 SIMDs; a wave64 vector instruction issues in 4 cycles (fp64 FMA: 78.6 TFLOP/s), v_mfma_f64_16x16x4_f64 in 64; 2.4 GHz nominal'}}
for f, name in (('4w', 'wallish2018'), ('4b', 'brieden2022')):
    own, other = collections.defaultdict(collections.Counter), collections.Counter()
    for path in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % f, recursive=True):
        for row in csv.DictReader(open(path)):
            kernel = row['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            if 'at::native' in kernel or kernel.startswith('__amd_rocclr'):
                other[row['Counter_Name']] += float(row['Counter_Value'])
            else:
                own[kernel][row['Counter_Name']] += float(row['Counter_Value'])
    line = None
    for l in open('/tmp/pmc_%s.log' % f):
        if l.startswith('{'):
            line = json.loads(l)
    nvec = line['vectors_through_the_filter_in_this_process']
    total = collections.Counter()
    for c in own.values():
        total.update(c)
    final[name] = {'vectors': nvec, 'chunk': line['chunk'], 'per_vector': {k: v / nvec for k, v in total.items()},
                   'per_vector_by_kernel': {k: {c: v / nvec for c, v in cnt.items()} for k, cnt in sorted(own.items(), key=lambda kv: -kv[1]['SQ_INSTS_VALU'])},
                   'framework_and_runtime_kernels_per_vector': {k: v / nvec for k, v in other.items()}}
json.dump(final, open('gpurun_out/%s_config4_valu.json' % tag, 'w'), indent=1)
print(json.dumps({k: final[k]['per_vector'] for k in ('wallish2018', 'brieden2022')}, indent=1))
PY
