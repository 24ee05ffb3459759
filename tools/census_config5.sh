#!/bin/bash
# vector wave-instructions per sample of config 5 (bg_kernel):   gpurun -- bash tools/census_config5.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/census_c5
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p -o c5 -- python3 $GRAFT_REPO_ROOT/tools/profile_secondary.py 5 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'bg_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
n = 1250000
out = {k: sum(v) / len(v) / n for k, v in acc.items()}
print(json.dumps({'samples_per_launch': n, 'per_sample': out, 'launches': {k: len(v) for k, v in acc.items()}}, indent=1))
import hashlib
lib = hashlib.sha256(open('$GRAFT_REPO_ROOT/cosmoprimo_amd/libcosmoprimo_amd.so', 'rb').read()).hexdigest()[:16]
json.dump({'library_sha256_16': lib, 'what': 'wave-instructions of bg_kernel<119, false, false> per (Omega_m, w0, wa, z) sample of config 5: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -- python3 tools/profile_secondary.py 5 (tools/census_config5.sh), mean over the launches of the run / 1 250 000 samples; multiply by 64 lanes for per-thread counts', 'samples_per_launch': n, 'per_sample': out}, open('$OUT/config5_valu.json', 'w'), indent=1)
PY
tail -1 $OUT/log.txt | cut -c1-400
